cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
for dt in bf16 bf16r; do
rm -rf $O/kst_$dt
DIC_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d $O/kst_$dt --output-format csv -- python $R/bench.py --quick --no-roofline --steps 10 --warmup 3 --dtype $dt > $O/kst_$dt.log 2>&1
f=$(find $O/kst_$dt -name "*kernel_stats.csv" | head -1); cp $f $O/r05b_${dt}_single_stream_kernel_stats.csv
grep -h '"metric"' $O/kst_$dt.log | cut -c1-200
rm -rf $O/kst_$dt
done
cd $R; timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -k "trained_collapsed" -s 2>&1 | grep -v "^$" | tail -22

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
rm -rf $O/kst_s
rocprofv3 --kernel-trace --stats -d $O/kst_s --output-format csv -- python $R/scripts/bench_sample.py --steps 100 --reps 1 --bleu-batch 0 > $O/kst_s.log 2>&1
f=$(find $O/kst_s -name "*kernel_stats.csv" | head -1); cp $f $O/r05a_sampling_kernel_stats.csv
tail -2 $O/kst_s.log | cut -c1-300
python $R/scripts/kstats.py $O/r05a_sampling_kernel_stats.csv | head -24
rm -rf $O/kst_s

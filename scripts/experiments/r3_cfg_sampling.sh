#!/bin/bash
# round 3: config 5 (seq 32 + guidance) and config 4 (sampling loop) -- tests, bench lines, per-kernel stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -k "timestep_embedding or config5 or golden_two_training or golden_eval or sampling or validate" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python bench.py --quick --no-roofline --seq-len 32 --cfg-weight 0.3 --steps 8 --warmup 3 2>/dev/null | tail -1 > $O/bench_seq32_cfg.json; cut -c1-200 $O/bench_seq32_cfg.json
kst() { tag=$1; shift; rm -rf $O/$tag; ( cd /tmp; rocprofv3 --kernel-trace --stats -d $O/$tag --output-format csv -- "$@" > $O/$tag.log 2>&1 )
  f=$(find $O/$tag -name "*kernel_stats.csv" | head -1); cp $f $O/${tag}_kernel_stats.csv; python scripts/kstats.py $f | head -${NSHOW:-24}; echo "ATen/copy kernels:"; grep -c "at::native\|rocclr" $f; }
kst seq32cfg python $R/bench.py --quick --no-roofline --no-cpu-baseline --seq-len 32 --cfg-weight 0.3 --steps 6 --warmup 2
kst sampling python $R/scripts/bench_sample.py --steps 20 --reps 1 --bleu-batch 0
DIC_SAMPLE_GRAPH=0 python scripts/bench_sample.py --bleu-batch 0 2>/dev/null | tail -1 | cut -c1-200
python scripts/bench_sample.py --bleu-batch 0 2>/dev/null | tail -1 | cut -c1-200

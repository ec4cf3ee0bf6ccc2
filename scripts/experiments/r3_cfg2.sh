#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3g; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python bench.py --quick --no-roofline --seq-len 32 --cfg-weight 0.3 --steps 8 --warmup 3 2>/dev/null | tail -1 > $O/bench_seq32_cfg.json; cut -c1-200 $O/bench_seq32_cfg.json
kst() { tag=$1; shift; rm -rf $O/$tag; ( cd /tmp; rocprofv3 --kernel-trace --stats -d $O/$tag --output-format csv -- "$@" > $O/$tag.log 2>&1 )
  f=$(find $O/$tag -name "*kernel_stats.csv" | head -1); cp $f $O/${tag}_kernel_stats.csv; python scripts/kstats.py $f | head -${NSHOW:-14}; echo "ATen/copy kernel names:"; grep -c "at::native\|rocclr" $f; rm -rf $O/$tag; }
kst seq32cfg python $R/bench.py --quick --no-roofline --no-cpu-baseline --seq-len 32 --cfg-weight 0.3 --steps 6 --warmup 2
NSHOW=8 kst sampling python $R/scripts/bench_sample.py --steps 20 --reps 1 --bleu-batch 0
python scripts/bench_sample.py --bleu-batch 0 2>/dev/null | tail -1 | cut -c1-200
(for tk in 18 34; do echo "Tk=$tk"; TK=$tk python scripts/attn_bench.py 2>&1 | grep p_drop; done)
python bench.py --quick 2>/dev/null | tail -1 | cut -c1-260

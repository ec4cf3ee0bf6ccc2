#!/bin/bash
# A/B: this round's gemm.hip (lock-step loop default) vs the round-2 gemm.hip, same everything else, interleaved, one box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3i; mkdir -p $O; cd $R
for i in 1 2 3; do
  python bench.py --quick --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('now ', d['value'], d['ms_per_step'])"
  DIC_HIP_LIB=ab/gemm_r02/libdic_hip.so python bench.py --quick --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r02 ', d['value'], d['ms_per_step'])"
done
COLD=1 TILE=256 python scripts/gemm_bench.py 2>&1 | grep TFLOP > $O/now.txt
DIC_HIP_LIB=ab/gemm_r02/libdic_hip.so COLD=1 TILE=256 python scripts/gemm_bench.py 2>&1 | grep TFLOP > $O/r02.txt
paste <(awk '{print $1,$2,$3,$4,$(NF-6),$(NF-4)}' $O/now.txt) <(awk '{print $(NF-6),$(NF-4)}' $O/r02.txt)

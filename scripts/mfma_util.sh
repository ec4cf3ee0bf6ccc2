#!/bin/bash
# MFMA utilisation of the training step per kernel (north_star: "evidenced by rocprof HBM GB/s and MFMA utilisation"): ONE rocprofv3 PMC pass
# (--kernel-trace + --pmc only; separate from the HBM passes of scripts/pmc_traffic.sh) over `bench.py --quick`, folded by mfma_util.py.
# usage (via gpurun): bash scripts/mfma_util.sh r03   -> gpurun_out/r03_mfma_util.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/mfma; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $O/p --output-format csv -- python $R/bench.py --quick --no-roofline --steps 3 --warmup 1 > $O/p.log 2>&1
C=$(find $O/p -name "*counter_collection.csv" | head -1); K=$(find $O/p -name "*kernel_trace.csv" | head -1)
python $R/scripts/mfma_util.py $C $K 4 > $R/gpurun_out/${TAG}_mfma_util.txt
head -40 $R/gpurun_out/${TAG}_mfma_util.txt

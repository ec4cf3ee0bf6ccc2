#!/bin/bash
# HBM traffic of the training step per kernel: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only -- see
# MI355X_MICROARCH.md, HBM section) over `bench.py --quick`, folded by pmc_traffic.py.  Run on the GPU box; writes gpurun_out/pmc/ and,
# with a tag argument (e.g. r02), profiles/<tag>_pmc_hbm_traffic.txt + profiles/<tag>_pmc_gemm_traffic.json (stamped with the csrc hash).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-}
STEPS=3; WARM=1
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc; rm -rf $O; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $O/$c --output-format csv -- python $R/bench.py --quick --no-roofline --steps $STEPS --warmup $WARM > $O/$c.log 2>&1
done
F=$(find $O/FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $O/WRITE_SIZE -name "*counter_collection.csv" | head -1)
SHA=$(cd $R && python -c "import bench; print(bench.csrc_sha())")
python $R/scripts/pmc_traffic.py $F $W $((STEPS + WARM)) $O/pmc_gemm_traffic.json $SHA > $O/pmc_hbm_traffic.txt
cat $O/pmc_hbm_traffic.txt | head -30
if [ -n "$TAG" ]; then cp $O/pmc_hbm_traffic.txt $R/gpurun_out/${TAG}_pmc_hbm_traffic.txt; cp $O/pmc_gemm_traffic.json $R/gpurun_out/${TAG}_pmc_gemm_traffic.json; fi

#!/bin/bash
# One gpurun call: the full GPU suite (log kept, stamped with the commit and the kernel-source hash), the round's evidence collection, the smoke entry.
# usage: gpurun --timeout 5400 -- 'bash scripts/gpu_round.sh r06 <git sha>'      (the .git directory does not travel: pass the sha)
TAG=${1:-rXX}; HEAD_SHA=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; O=$R/gpurun_out; mkdir -p $O
CSRC=$(python -c "import bench; print(bench.csrc_sha())")
LOG=$O/${TAG}_pytest_gpu.txt
{ echo "# python -m pytest tests -m gpu -q    HEAD $HEAD_SHA   csrc_sha $CSRC   $(date -u +%FT%TZ)"; rocm-smi --showproductname 2>/dev/null | grep -m1 -i "card series"; } > $LOG
timeout 3000 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids >> $LOG
echo "# pytest exit code ${PIPESTATUS[0]}" >> $LOG
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/${TAG}_smoke.txt 2>&1
[ "$3" = "nocollect" ] || bash scripts/collect_profiles.sh $TAG > $O/collect.log 2>&1
tail -5 $LOG; tail -3 $O/${TAG}_smoke.txt; cut -c1-600 $O/${TAG}_bench.json 2>/dev/null

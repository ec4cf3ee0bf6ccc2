#!/bin/bash
# scratch driver for one gpurun call: the full GPU suite, the round's evidence collection, the smoke entry
cd /root/repo
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/pytest_gpu.txt
bash scripts/collect_profiles.sh r05 > gpurun_out/collect.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/smoke.txt 2>&1
cat gpurun_out/pytest_gpu.txt; tail -3 gpurun_out/smoke.txt; cut -c1-400 gpurun_out/r05_bench.json

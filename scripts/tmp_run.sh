#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q -k "four_wave" 2>&1 | tail -4 > gpurun_out/w4ag_pytest.txt
for i in 1 2; do
  python bench.py --mode sample 2>&1 | tail -1 | cut -c1-300 > gpurun_out/sampleg_mask073_$i.json
  DIC_OPTIONS=gemm_w4a_mask=0x173 python bench.py --mode sample 2>&1 | tail -1 | cut -c1-300 > gpurun_out/sampleg_mask173_$i.json
  python bench.py --quick --no-roofline --steps 40 2>&1 | tail -1 | cut -c1-200 > gpurun_out/traing_mask073_$i.json
  DIC_OPTIONS=gemm_w4a_mask=0x273 python bench.py --quick --no-roofline --steps 40 2>&1 | tail -1 | cut -c1-200 > gpurun_out/traing_mask273_$i.json
  DIC_OPTIONS=gemm_w4a_mask=0x07b python bench.py --quick --no-roofline --steps 40 2>&1 | tail -1 | cut -c1-200 > gpurun_out/traing_mask07b_$i.json
done
python scripts/gemm_in_step.py > gpurun_out/gis_073.txt 2>&1
DIC_OPTIONS=gemm_w4a_mask=0x273 python scripts/gemm_in_step.py > gpurun_out/gis_273.txt 2>&1
cat gpurun_out/w4ag_pytest.txt; cat gpurun_out/sampleg_*.json gpurun_out/traing_*.json; cat gpurun_out/gis_073.txt gpurun_out/gis_273.txt

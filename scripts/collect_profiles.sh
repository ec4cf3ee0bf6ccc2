#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/<tag>_*; copy what should be judged into profiles/ afterwards.
# usage (via gpurun): bash scripts/collect_profiles.sh r02
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; O=$R/gpurun_out; mkdir -p $O
# PMC traffic first: bench.py quotes roofline.traffic only from a collection stamped with the hash of the csrc/ it runs on
bash scripts/pmc_traffic.sh $TAG > /dev/null 2>&1
cp $O/${TAG}_pmc_gemm_traffic.json profiles/ 2>/dev/null
python bench.py 2>/dev/null | tail -1 > $O/${TAG}_bench.json
python bench.py --quick --layers 6 2>/dev/null | tail -1 > $O/${TAG}_bench_6layer.json
python bench.py --quick --layers 6 --batch 8 --sample-size 100 2>/dev/null | tail -1 > $O/${TAG}_bench_S100_B8_6layer.json
python bench.py --mode sample 2>/dev/null | tail -1 > $O/${TAG}_sampling_config4.json
python bench.py --quick --seq-len 32 --cfg-weight 0.3 2>/dev/null | tail -1 > $O/${TAG}_bench_seq32_cfg.json
bash scripts/experiments/step_kstats.sh > $O/${TAG}_kstats_two_streams.txt 2>&1; cp $O/kstats/new_kernel_stats.csv $O/${TAG}_bench_kernel_stats.csv
WGS=0 bash scripts/experiments/step_kstats.sh > $O/${TAG}_kstats_single_stream.txt 2>&1; cp $O/kstats/new_kernel_stats.csv $O/${TAG}_single_stream_kernel_stats.csv
TILE=256 python scripts/gemm_bench.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_gemm_microbench.txt
COLD=1 TILE=256 python scripts/gemm_bench.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_gemm_microbench_cold.txt
(for tk in 18 34; do echo "Tk=$tk"; TK=$tk python scripts/attn_bench.py 2>&1 | grep p_drop; done) > $O/${TAG}_attn_microbench.txt
if [ -f ab/libdic_trace.so ]; then DIC_HIP_LIB=$R/ab/libdic_trace.so python scripts/experiments/gemm_trace.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_gemm_phase_trace.txt; fi
ls -la $O/${TAG}_*

#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/<tag>_*; copy what should be judged into profiles/ afterwards.
# usage (via gpurun): bash scripts/collect_profiles.sh r03
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
f=$(find $O/kstats/new -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python scripts/gap_analysis.py $f > $O/${TAG}_gap_analysis.txt 2>&1
WGS=0 bash scripts/experiments/step_kstats.sh > $O/${TAG}_kstats_single_stream.txt 2>&1; cp $O/kstats/new_kernel_stats.csv $O/${TAG}_single_stream_kernel_stats.csv
# MFMA utilisation (its own PMC pass), config 5 and config 4 per-kernel summaries
bash scripts/mfma_util.sh $TAG > /dev/null 2>&1
kst() { tag=$1; shift; rm -rf $O/kst_$tag; ( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats -d $O/kst_$tag --output-format csv -- "$@" > $O/kst_$tag.log 2>&1 )
  f=$(find $O/kst_$tag -name "*kernel_stats.csv" | head -1); cp $f $O/${TAG}_${tag}_kernel_stats.csv; rm -rf $O/kst_$tag; }
kst seq32_cfg python $R/bench.py --quick --no-roofline --seq-len 32 --cfg-weight 0.3 --steps 10 --warmup 3
DIC_WGRAD_STREAM=0 kst seq32_cfg_single_stream python $R/bench.py --quick --no-roofline --seq-len 32 --cfg-weight 0.3 --steps 10 --warmup 3
kst sampling python $R/scripts/bench_sample.py --steps 100 --reps 1 --bleu-batch 0
TILE=256 python scripts/gemm_bench.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_gemm_microbench.txt
COLD=1 TILE=256 python scripts/gemm_bench.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_gemm_microbench_cold.txt
(for tk in 18 34; do echo "Tk=$tk"; TK=$tk python scripts/attn_bench.py 2>&1 | grep p_drop; done) > $O/${TAG}_attn_microbench.txt
python scripts/gemm_in_step.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_gemm_in_step_bf16.txt
python scripts/gemm_in_step.py --dtype bf16r 2>&1 | grep -v amdgpu.ids > $O/${TAG}_gemm_in_step_bf16r.txt
python scripts/gemm_in_step.py --dtype bf16w 2>&1 | grep -v amdgpu.ids > $O/${TAG}_gemm_in_step_bf16w.txt
timeout 300 python scripts/experiments/w4a_check.py time 2>&1 | grep -v amdgpu.ids > $O/${TAG}_gemm_w4a_check.txt
timeout 300 python scripts/experiments/tile_rows_probe.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_tile_rows_probe.txt
[ -f abl/libdic_clk.so ] && timeout 900 python scripts/power_ab.py --steps 400 --pin 1900 > $O/${TAG}_power_ab.txt 2>&1
python bench.py --quick --dtype bf16w 2>/dev/null | tail -1 > $O/${TAG}_bench_bf16w.json
python bench.py --quick --dtype bf16r 2>/dev/null | tail -1 > $O/${TAG}_bench_bf16r.json
timeout 900 python scripts/experiments/mode_trajectory_probe.py --time 2>&1 | grep -v amdgpu.ids > $O/${TAG}_mode_trajectory.txt
# loss distance of the bf16 engines from fp32 ALONG a training run (which lo halves, which rounding points): DESIGN.md section 4
if [ -f ab/libdic_trace.so ]; then DIC_HIP_LIB=$R/ab/libdic_trace.so python scripts/experiments/gemm_trace.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_gemm_phase_trace.txt; fi
ls -la $O/${TAG}_*

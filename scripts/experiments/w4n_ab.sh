#!/bin/bash
# First hardware run of the NARROW-tile asm GEMM (csrc/gemm_w4n.h, option gemm_w4n; built and proven on the CPU only so far), on the GPU box:
#   bash scripts/experiments/w4n_ab.sh            (through gpurun; ~12 minutes)
# 1. parity: the narrow bodies against the wide ones bit for bit + float64 (tests/test_gpu_zz_w4n.py with --runxfail: 133 cases);
# 2. per launch inside the step (scripts/gemm_in_step.py), option off / on;
# 3. the step and the sampling pass, interleaved off / on (on = the default mask 0x740: the heavy-epilogue forms), twice; then per (layout, epilogue) form: which forms pay (gemm_w4n_mask one bit at a time).
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R; O=$R/gpurun_out; mkdir -p $O
{
echo "# narrow-tile asm GEMM: first hardware run   $(date -u +%FT%TZ)"
timeout 1500 python -m pytest tests/test_gpu_zz_w4n.py -q --runxfail -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -6
echo "## GEMM launches inside the step, gemm_w4n=0"; python scripts/gemm_in_step.py 2>&1 | grep -v amdgpu.ids
echo "## GEMM launches inside the step, gemm_w4n=1"; DIC_OPTIONS=gemm_w4n=1 python scripts/gemm_in_step.py 2>&1 | grep -v amdgpu.ids
for i in 1 2; do
  echo "## step, wide:    $(python bench.py --quick --no-roofline --steps 40 2>/dev/null | tail -1 | cut -c1-140)"
  echo "## step, narrow:  $(DIC_OPTIONS=gemm_w4n=1 python bench.py --quick --no-roofline --steps 40 2>/dev/null | tail -1 | cut -c1-140)"
  echo "## pass, wide:    $(python bench.py --mode sample 2>/dev/null | tail -1 | cut -c1-170)"
  echo "## pass, narrow:  $(DIC_OPTIONS=gemm_w4n=1 python bench.py --mode sample 2>/dev/null | tail -1 | cut -c1-170)"
done
echo "## step, narrow on EVERY eligible form (gemm_w4n_mask=0x7ff):  $(DIC_OPTIONS=gemm_w4n=1,gemm_w4n_mask=0x7ff python bench.py --quick --no-roofline --steps 40 2>/dev/null | tail -1 | cut -c1-140)"
echo "## pass, narrow on EVERY eligible form (gemm_w4n_mask=0x7ff):  $(DIC_OPTIONS=gemm_w4n=1,gemm_w4n_mask=0x7ff python bench.py --mode sample 2>/dev/null | tail -1 | cut -c1-170)"
echo "## step, narrow, loop form everywhere (gemm_w4n_flat=0):  $(DIC_OPTIONS=gemm_w4n=1,gemm_w4n_flat=0 python bench.py --quick --no-roofline --steps 40 2>/dev/null | tail -1 | cut -c1-140)"
echo "## pass, narrow, loop form everywhere (gemm_w4n_flat=0):  $(DIC_OPTIONS=gemm_w4n=1,gemm_w4n_flat=0 python bench.py --mode sample 2>/dev/null | tail -1 | cut -c1-170)"
# which forms pay: bit 4 * b_km + v (v = 0 plain, 1 + residual, 2 x aux, 3 dropout + residual), bit 8 GELU, bit 9 GELU + GELU', bit 10 the rounding-head forward (CE_EXP)
for bit in 0 1 2 3 4 5 6 8 9 10; do
  m=$((1 << bit))
  echo "## step, narrow only for mask bit $bit:  $(DIC_OPTIONS=gemm_w4n=1,gemm_w4n_mask=$m,gemm_w4a_mask=0x3ff python bench.py --quick --no-roofline --steps 40 2>/dev/null | tail -1 | cut -c1-140)"
done
# generator options of the narrow bodies (variant libraries built by scripts/experiments/build_ab_libs.sh from the same sources)
if [ "$(cat abl/BUILT_FROM 2>/dev/null)" = "$(python -c 'import bench; print(bench.csrc_sha())')" ]; then
  for v in w4n_pk1 w4n_bar2 w4n_bar4 w4n_quota4; do
    [ -f abl/libdic_$v.so ] || continue
    echo "## step, narrow, variant $v:  $(DIC_HIP_LIB=$R/abl/libdic_$v.so DIC_OPTIONS=gemm_w4n=1 python bench.py --quick --no-roofline --steps 40 2>/dev/null | tail -1 | cut -c1-140)"
    echo "## pass, narrow, variant $v:  $(DIC_HIP_LIB=$R/abl/libdic_$v.so DIC_OPTIONS=gemm_w4n=1 python bench.py --mode sample 2>/dev/null | tail -1 | cut -c1-170)"
  done
  echo "## step, narrow, shipped again:  $(DIC_OPTIONS=gemm_w4n=1 python bench.py --quick --no-roofline --steps 40 2>/dev/null | tail -1 | cut -c1-140)"
else
  echo "## variant libraries are missing or were built from other sources (bash scripts/experiments/build_ab_libs.sh): skipped"
fi
echo "## step, wide with every form on the asm kernel (mask 0x3ff): $(DIC_OPTIONS=gemm_w4a_mask=0x3ff python bench.py --quick --no-roofline --steps 40 2>/dev/null | tail -1 | cut -c1-140)"
} > $O/r06_w4n_ab.txt 2>&1
cat $O/r06_w4n_ab.txt

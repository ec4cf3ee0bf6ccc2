#!/bin/bash
# A/B of generator options of the four-wave asm GEMM (scripts/gen_w4a.py OUT <options>) against the shipped bodies, on the GPU box:
#   bash scripts/experiments/w4a_variant_ab.sh block_waits early_side defer_stores       (any subset; run through gpurun; ~8 minutes)
# The variant text is first proven on the CPU (race checker + emulator), then built into abl/libdic_w4avar.so (-DW4A_ASM_INC), checked against the
# 8-wave kernel on the GPU (w4a_check.py) and timed: per shape (tile_rows_probe.py) and in the step / the sampling pass, interleaved with the shipped library.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R; O=$R/gpurun_out; mkdir -p $O abl
OPTS="$@"; TAG=$(echo $OPTS | tr ' ' '_')
python scripts/gen_w4a.py abl/w4a_var.inc $OPTS 2> /dev/null
python - <<PY
import sys
sys.path.insert(0, "scripts")
import w4a_hazard_check as H, w4a_emulate as W
opts = tuple("$OPTS".split())
print("race checker:", H.check_all(opts=opts), "runs ok")
bad = sum((lambda r: r[0] > 1.0 or not r[1])(W.run_case(ni, bkm, epi, 32 * ni + 80, 512, 384, opts=opts)) for ni, bkm, epi in W.bodies())
print("emulator:", "all bodies reproduce numpy" if not bad else f"{bad} bodies DIFFER")
sys.exit(1 if bad else 0)
PY
# (the variant library is normally built in the CPU container by scripts/experiments/build_ab_libs.sh and travels with the snapshot: rebuild only if it is missing, was built
#  from other kernel sources or for other options -- 2.5 minutes of GPU-box time otherwise)
if [ -f abl/libdic_w4avar.so ] && [ "$(cat abl/BUILT_FROM 2>/dev/null)" = "$(python -c 'import bench; print(bench.csrc_sha())')" ] && [ "$(cat abl/w4a_var.opts 2>/dev/null)" = "$OPTS" ]; then
  echo "variant library: reusing abl/libdic_w4avar.so"
else
  bash scripts/build_variant.sh w4avar "-DW4A_ASM_INC=<w4a_var.inc> -I$R/abl" > /dev/null && echo "$OPTS" > abl/w4a_var.opts
fi
{
echo "# asm GEMM generator options: $OPTS"
DIC_HIP_LIB=$R/abl/libdic_w4avar.so python scripts/experiments/w4a_check.py 2>&1 | tail -1
echo "## per shape, shipped library"; python scripts/experiments/tile_rows_probe.py 2>&1 | grep -v amdgpu.ids
echo "## per shape, variant"; DIC_HIP_LIB=$R/abl/libdic_w4avar.so python scripts/experiments/tile_rows_probe.py 2>&1 | grep -v amdgpu.ids
for i in 1 2; do
  echo "## step, shipped:  $(python bench.py --quick --no-roofline --steps 40 2>/dev/null | tail -1 | cut -c1-140)"
  echo "## step, variant:  $(DIC_HIP_LIB=$R/abl/libdic_w4avar.so python bench.py --quick --no-roofline --steps 40 2>/dev/null | tail -1 | cut -c1-140)"
  echo "## pass, shipped:  $(python bench.py --mode sample 2>/dev/null | tail -1 | cut -c1-170)"
  echo "## pass, variant:  $(DIC_HIP_LIB=$R/abl/libdic_w4avar.so python bench.py --mode sample 2>/dev/null | tail -1 | cut -c1-170)"
done
} > $O/w4a_variant_${TAG}.txt 2>&1
cat $O/w4a_variant_${TAG}.txt

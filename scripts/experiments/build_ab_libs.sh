#!/bin/bash
# Builds, HERE (hipcc cross-compiles; abl/ travels to the GPU box with the snapshot, is not committed), every variant library the round's A/B scripts load:
#   clk       -DDIC_CLOCK_PROBE                         scripts/power_ab.py
#   w4avar    the wide asm GEMM with block_waits early_side defer_stores      scripts/experiments/w4a_variant_ab.sh (rebuilds it itself when asked for other options)
#   w4n_bar2 / w4n_bar4 / w4n_quota4 / w4n_pk1          generator options of the narrow asm GEMM (scripts/gen_w4n.py)     scripts/experiments/w4n_ab.sh
# Run after every change to csrc/ (a variant library built from older sources would be A/B-ing two things at once).   ~8 minutes on 8 cores.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd); cd $R; mkdir -p abl
python scripts/gen_w4a.py abl/w4a_var.inc block_waits early_side defer_stores 2> /dev/null
echo "block_waits early_side defer_stores" > abl/w4a_var.opts
python scripts/gen_w4n.py abl/w4n_bar2.inc bar=2 2> /dev/null
python scripts/gen_w4n.py abl/w4n_bar4.inc bar=4 2> /dev/null
python scripts/gen_w4n.py abl/w4n_quota4.inc quota=4 2> /dev/null
python scripts/gen_w4n.py abl/w4n_pk1.inc pk=1 2> /dev/null
( bash scripts/build_variant.sh clk "-DDIC_CLOCK_PROBE" > /dev/null; bash scripts/build_variant.sh w4avar "-DW4A_ASM_INC=<w4a_var.inc> -I$R/abl" > /dev/null ) &
( bash scripts/build_variant.sh w4n_bar2 "-DW4N_ASM_INC=<w4n_bar2.inc> -I$R/abl" > /dev/null; bash scripts/build_variant.sh w4n_bar4 "-DW4N_ASM_INC=<w4n_bar4.inc> -I$R/abl" > /dev/null ) &
( bash scripts/build_variant.sh w4n_quota4 "-DW4N_ASM_INC=<w4n_quota4.inc> -I$R/abl" > /dev/null; bash scripts/build_variant.sh w4n_pk1 "-DW4N_ASM_INC=<w4n_pk1.inc> -I$R/abl" > /dev/null ) &
wait
python - <<'PY'
import hashlib, glob, os, sys
sys.path.insert(0, ".")
import bench
open("abl/BUILT_FROM", "w").write(bench.csrc_sha() + "\n")
print("variant libraries built from csrc", bench.csrc_sha())
PY
ls -la abl/*.so

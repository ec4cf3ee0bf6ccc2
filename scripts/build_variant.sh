#!/bin/bash
# Build another copy of the HIP library with extra -D flags for within-run A/B measurements (load it with DIC_HIP_LIB=<path>):
#   scripts/build_variant.sh NAME "-DDIC_GEMM_WIDE_ORDER=0 -DDIC_CE_EXP_NT=0"   ->  abl/libdic_NAME.so   (abl/ travels to the GPU box, is not committed)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FLAGS=$2
mkdir -p $R/abl/obj_$NAME
C=$R/diffusion-image-captioning_amd/csrc
pids=""
for f in gemm attn norm misc; do
  extra=""; [ $f = misc ] && extra="-ffp-contract=off"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $extra -std=c++20 -fPIC -Wno-unused-value $FLAGS -c $C/$f.hip -o $R/abl/obj_$NAME/$f.o 2> $R/abl/obj_$NAME/$f.log &
  pids="$pids $!"
done
for p in $pids; do wait $p || { echo "build_variant.sh $NAME: a compile FAILED:"; grep -h -A3 "error" $R/abl/obj_$NAME/*.log | head -20; exit 1; }; done      # (a bare `wait` hides a failed compile: round 6 found a 0.5 MB "library" without its GEMMs)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/abl/libdic_$NAME.so $R/abl/obj_$NAME/*.o
rm -rf $R/abl/obj_$NAME
ls -la $R/abl/libdic_$NAME.so

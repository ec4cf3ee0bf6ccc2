#!/bin/bash
# Build another copy of the HIP library with extra -D flags for within-run A/B measurements (load it with DIC_HIP_LIB=<path>):
#   scripts/build_variant.sh NAME "-DDIC_GEMM_WIDE_ORDER=0 -DDIC_CE_EXP_NT=0"   ->  abl/libdic_NAME.so   (abl/ travels to the GPU box, is not committed)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FLAGS=$2
mkdir -p $R/abl/obj_$NAME
C=$R/diffusion-image-captioning_amd/csrc
for f in gemm attn norm misc; do
  extra=""; [ $f = misc ] && extra="-ffp-contract=off"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $extra -std=c++20 -fPIC -Wno-unused-value $FLAGS -c $C/$f.hip -o $R/abl/obj_$NAME/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/abl/libdic_$NAME.so $R/abl/obj_$NAME/*.o
rm -rf $R/abl/obj_$NAME
ls -la $R/abl/libdic_$NAME.so

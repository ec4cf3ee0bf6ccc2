#!/bin/bash
# Timing ablations of the CE_EXP epilogue: builds libdic_hip.so variants into ab/ce_<name>/ (here, on the build box; they travel with the snapshot),
# then `python scripts/experiments/ce_exp_ablate.py` times them on the GPU.  Usage: scripts/experiments/ce_exp_ablate.sh base:"" nosum:"-DDIC_CE_EXP_ABL=1" ...
set -e
cd "$(dirname "$0")/../.."
SRC=diffusion-image-captioning_amd/csrc
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  mkdir -p ab/ce_$name
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wno-unused-value $flags -c $SRC/gemm.hip -o ab/ce_$name/gemm.o \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/ce_$name/libdic_hip.so ab/ce_$name/gemm.o $SRC/attn.o $SRC/norm.o $SRC/misc.o \
    && echo built $name ) &
done
wait

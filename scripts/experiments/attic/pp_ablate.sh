#!/bin/bash
# Timing ablations of the ping-pong GEMM loop (csrc/gemm_pp.h): builds libdic_hip.so variants with one ingredient removed each into ab/pp_<name>/
# (run here, on the build box; the variants travel to the GPU box with the snapshot), then `scripts/experiments/pp_ablate.py` times them.
# Usage: scripts/experiments/pp_ablate.sh name1:"-DFLAG1 -DFLAG2" name2:"..."
set -e
cd "$(dirname "$0")/../.."
SRC=diffusion-image-captioning_amd/csrc
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  mkdir -p ab/pp_$name
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wno-unused-value -DDIC_GEMM_MIN $flags -c $SRC/gemm.hip -o ab/pp_$name/gemm.o \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/pp_$name/libdic_hip.so ab/pp_$name/gemm.o $SRC/attn.o $SRC/norm.o $SRC/misc.o \
    && echo built $name ) &
done
wait

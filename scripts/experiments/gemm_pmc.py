#!/usr/bin/env python3
"""A handful of GEMM shapes, few launches each, for `rocprofv3 --pmc` passes (SQ counters per kernel launch).
usage: rocprofv3 --kernel-trace --pmc <counters> -d <dir> --output-format csv -- python scripts/experiments/gemm_pmc.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gemm_bench as gb
T, D, F = 17408, 768, 3072
for tile in ("256", "128"):
    os.environ["TILE"] = tile
    print("TILE", tile)
    gb.run("square 4096", 4096, 4096, 4096, 0, 0, iters=3)
    gb.run("fwd qkv", T, 3 * D, D, 0, 0, iters=3)
    gb.run("fwd ffn1 gelu", T, F, D, 0, 0, epi=1, iters=3)
    gb.run("dX gelu'", T, F, D, 0, 1, epi=2, iters=3)
    gb.run("dX qkv->dh", T, D, 3 * D, 0, 1, iters=3)

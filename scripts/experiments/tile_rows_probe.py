#!/usr/bin/env python3
"""Does a tile's time scale with its HEIGHT?  (round 5: is a 224-row variant of the four-wave asm GEMM worth generating?)

For the step's and the sampling pass's forward / input-gradient shapes, cold operands, three kernels per shape:
  8-wave kernel with full-height (256-row) tiles only   (dic_set_option gemm_rows=0)
  8-wave kernel with the per-launch tile height          (gemm_rows=1: 224 rows at 17 408 tokens)
  four-wave asm kernel with 256-row tiles, with 224-row tiles (round 5: scripts/gen_w4a.py ni = 7), and with the height launch_w4a picks
With one round of tiles either way (17 408 x 768: 204 vs 234 tiles) the ratio of the first two IS the per-tile time ratio.
"""
import ctypes as C
import importlib
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dic = importlib.import_module("diffusion-image-captioning_amd")
L = dic.lib()
GP = dic._lib.GemmParams
bf = torch.bfloat16


def sets_for(M, N, K, b_km, resid):
    per = (M * K + N * K + M * N * (2 if resid else 1)) * 2
    nset = max(3, min(16, int(1.2e9 // per)))
    out = []
    for _ in range(nset):
        A = torch.randn(M, K, device="cuda").to(bf)
        B = torch.randn((K, N) if b_km else (N, K), device="cuda").to(bf)
        Cc = torch.empty(M, N, device="cuda", dtype=bf)
        R = torch.randn(M, N, device="cuda").to(bf) if resid else None
        bias = torch.randn(N, device="cuda")
        g = GP(A=A.data_ptr(), B=B.data_ptr(), C=Cc.data_ptr(), M=M, N=N, K=K, lda=K, ldb=B.shape[1], ldc=N, tile=256, bias=bias.data_ptr(),
               R=R.data_ptr() if resid else 0, ldr=N, split_k=1, seed=7)
        out.append((g, A, B, Cc, R, bias))
    return out


def timeit(sets, b_km, iters=24):
    st = torch.cuda.current_stream().cuda_stream
    for g, *_ in sets:
        assert L.dic_gemm(1, 0, b_km, 0, C.byref(g), st) == 0, L.dic_last_error()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = max(iters, len(sets))
    e0.record()
    for i in range(n):
        L.dic_gemm(1, 0, b_km, 0, C.byref(sets[i % len(sets)][0]), st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    print("# us per launch, cold operands; tiles = (M / rows) x (N / 256) on 256 CUs")
    print(f"# {'shape':64s} {'8-wave 256':>11s} {'8-wave auto':>12s} {'asm 256':>9s} {'asm 224':>9s} {'asm auto':>9s}")
    for M in (17408, 34816):
        for name, N, K, b_km, resid in (("out-proj fwd + residual", 768, 768, 0, True), ("FFN lin2 fwd + residual", 768, 3072, 0, True), ("q|k|v fwd", 2304, 768, 0, False),
                                        ("dX of q|k|v (k-major B) + residual", 768, 2304, 1, True), ("dX of FFN lin1 (k-major B) + residual", 768, 3072, 1, True)):
            sets = sets_for(M, N, K, b_km, resid)
            res = []
            for w4a, rows, arows in ((0, 0, 0), (0, 1, 0), (1, 1, 256), (1, 1, 224), (1, 1, 0)):
                assert L.dic_set_option(b"gemm_w4a", w4a) == 0 and L.dic_set_option(b"gemm_w4a_mask", 0xFF) == 0 and L.dic_set_option(b"gemm_rows", rows) == 0
                assert L.dic_set_option(b"gemm_w4a_rows", arows) == 0
                res.append(min(timeit(sets, b_km) for _ in range(3)))
            print(f"  M={M:6d} N={N:5d} K={K:5d} {name:38s} {res[0]:11.1f} {res[1]:12.1f} {res[2]:9.1f} {res[3]:9.1f} {res[4]:9.1f}", flush=True)
            del sets
            torch.cuda.empty_cache()
    L.dic_set_option(b"gemm_rows", 1)
    L.dic_set_option(b"gemm_w4a_rows", 0)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""GEMM microbenchmark on the step's shapes (run on the GPU box): TFLOP/s per (layout, epilogue, shape), v1 vs v2."""
import ctypes as C
import importlib
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dic = importlib.import_module("diffusion-image-captioning_amd")
L = dic.lib()
GP = dic._lib.GemmParams
T, D, F, V = int(os.environ.get("TOKENS", "17408")), 768, 3072, 30592
bf = torch.bfloat16


COLD = os.environ.get("COLD", "0") in ("1", "in", "out")   # rotate through enough operand/output sets to defeat the 256 MB infinity cache
COLD_MODE = os.environ.get("COLD", "0")                 # "in": only A/B rotate; "out": only C/aux/R rotate


def run(name, M, N, K, a_km, b_km, epi=0, split=1, out_f32=0, iters=20, resid=False, p_drop=0.0, **extra):
    if COLD:
        return run_cold(name, M, N, K, a_km, b_km, epi, split, out_f32, iters, resid, p_drop)
    A = torch.randn((K, M) if a_km else (M, K), device="cuda").to(bf)
    B = torch.randn((K, N) if b_km else (N, K), device="cuda").to(bf)
    Cc = torch.empty(M, N, device="cuda", dtype=torch.float32 if out_f32 else bf)
    aux = torch.randn(M, N, device="cuda").to(bf) if epi in (1, 2, 6, 7) else None
    bias = torch.randn(N, device="cuda") if epi in (0, 1, 6) and not out_f32 else None
    ws = torch.empty(split * M * N, device="cuda") if split > 1 else None
    Rr = torch.randn(M, N, device="cuda").to(bf) if resid else None
    g = GP(A=A.data_ptr(), B=B.data_ptr(), C=Cc.data_ptr(), M=M, N=N, K=K, lda=A.shape[1], ldb=B.shape[1], ldc=N, tile=int(os.environ.get("TILE", "128")),
           bias=bias.data_ptr() if bias is not None else 0, aux=aux.data_ptr() if aux is not None else 0, ldaux=N, out_f32=out_f32,
           split_k=split, split_ws=ws.data_ptr() if ws is not None else 0, R=Rr.data_ptr() if resid else 0, ldr=N, p_drop=p_drop, seed=7)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        assert L.dic_gemm(1, a_km, b_km, epi, C.byref(g), st) == 0, L.dic_last_error()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        L.dic_gemm(1, a_km, b_km, epi, C.byref(g), st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:34s} M={M:6d} N={N:6d} K={K:6d} split={split:2d}  {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)


def run_cold(name, M, N, K, a_km, b_km, epi, split, out_f32, iters, resid, p_drop):
    per = (M * K + N * K) * 2 + M * N * (4 if out_f32 else 2) * (2 if epi in (1, 2, 6, 7) else 1)
    nset = max(2, min(24, int(1.5e9 // per)))
    sets = []
    for _ in range(nset):
        A = torch.randn((K, M) if a_km else (M, K), device="cuda").to(bf)
        B = torch.randn((K, N) if b_km else (N, K), device="cuda").to(bf)
        Cc = torch.empty(M, N, device="cuda", dtype=torch.float32 if out_f32 else bf)
        aux = torch.randn(M, N, device="cuda").to(bf) if epi in (1, 2, 6, 7) else None
        bias = torch.randn(N, device="cuda") if epi in (0, 1, 6) and not out_f32 else None
        Rr = torch.randn(M, N, device="cuda").to(bf) if resid else None
        ws = torch.empty(split * M * N, device="cuda") if split > 1 else None
        g = GP(A=A.data_ptr(), B=B.data_ptr(), C=Cc.data_ptr(), M=M, N=N, K=K, lda=A.shape[1], ldb=B.shape[1], ldc=N, tile=int(os.environ.get("TILE", "128")),
               bias=bias.data_ptr() if bias is not None else 0, aux=aux.data_ptr() if aux is not None else 0, ldaux=N, out_f32=out_f32,
               split_k=split, split_ws=ws.data_ptr() if ws is not None else 0, R=Rr.data_ptr() if resid else 0, ldr=N, p_drop=p_drop, seed=7)
        if sets and COLD_MODE == "in":      # same outputs every launch
            g.C, g.aux, g.R = sets[0][0].C, sets[0][0].aux, sets[0][0].R
        if sets and COLD_MODE == "out":     # same operands every launch
            g.A, g.B = sets[0][0].A, sets[0][0].B
        sets.append((g, A, B, Cc, aux, bias, Rr, ws))
    st = torch.cuda.current_stream().cuda_stream
    for g, *_ in sets:
        assert L.dic_gemm(1, a_km, b_km, epi, C.byref(g), st) == 0, L.dic_last_error()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = max(iters, nset)
    e0.record()
    for i in range(n):
        L.dic_gemm(1, a_km, b_km, epi, C.byref(sets[i % nset][0]), st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name:34s} M={M:6d} N={N:6d} K={K:6d} split={split:2d}  {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s  (cold, {nset} sets)", flush=True)


if __name__ == "__main__" and len(sys.argv) == 1:
    print("TILE =", os.environ.get("TILE", "128"))
    run("fwd qkv        (KC,KC) bias", T, 3 * D, D, 0, 0)
    run("fwd out-proj   (KC,KC) bias", T, D, D, 0, 0)
    run("fwd out-proj   +resid +dropout", T, D, D, 0, 0, resid=True, p_drop=0.1)
    run("fwd ffn1       (KC,KC) gelu", T, F, D, 0, 0, epi=1)
    run("fwd ffn2       (KC,KC) bias", T, D, F, 0, 0)
    run("fwd ffn2       +resid +dropout", T, D, F, 0, 0, resid=True, p_drop=0.1)
    run("dX  ffn2->du   (KC,KM) gelu'", T, F, D, 0, 1, epi=2)
    run("dX  ffn1->dsa  (KC,KM)", T, D, F, 0, 1)
    run("dX  qkv->dh    (KC,KM)", T, D, 3 * D, 0, 1)
    for sp in (1, 4, 8, 14, 28):
        run("dW  out-proj   (KM,KM) f32", D, D, T, 1, 1, split=sp, out_f32=1)
    for sp in (1, 3, 4, 7):
        run("dW  ffn1       (KM,KM) f32", F, D, T, 1, 1, split=sp, out_f32=1)
    for sp in (1, 4, 9):
        run("dW  qkv        (KM,KM) f32", 3 * D, D, T, 1, 1, split=sp, out_f32=1)
    run("rounding dX    (KC,KM) f32", 16384, D, V, 0, 1, out_f32=1)
    run("square 4096    (KC,KC)", 4096, 4096, 4096, 0, 0)
    run("square 8192    (KC,KC)", 8192, 8192, 8192, 0, 0, iters=5)


def sweep():
    print("--- K sweep (KC,KC) bias epilogue, N=768")
    for M in (128 * 85, 18432):
        for K in (64, 256, 768, 1536, 3072, 6144):
            run(f"M={M} K sweep", M, 768, K, 0, 0)
    print("--- N=3072")
    for K in (64, 768, 3072):
        run("N=3072 K sweep", 18432, 3072, K, 0, 0)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "sweep":
    sweep()


def grouped_potential():
    """What one launch covering a layer's four weight gradients could reach: same tile count as a single (9216 x 768) x 18432 problem."""
    print("--- aggregated dW shape (2304+768+3072+3072) x 768, K=18432")
    for sp in (1, 2, 3, 4, 7):
        run("dW  layer-aggregate (KM,KM) f32", 9216, D, T, 1, 1, split=sp, out_f32=1)
    print("--- today: four launches")
    run("dW  qkv", 3 * D, D, T, 1, 1, split=9, out_f32=1)
    run("dW  out", D, D, T, 1, 1, split=14, out_f32=1)
    run("dW  ffn1", F, D, T, 1, 1, split=7, out_f32=1)
    run("dW  ffn2", D, F, T, 1, 1, split=7, out_f32=1)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "grouped":
    grouped_potential()


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "ce":
    for sp in (1, 2, 3, 4, 6):
        run("rounding dX    (KC,KM) f32", 16384, D, V, 0, 1, out_f32=1, split=sp)
    run("rounding logits(KC,KC) plain", 16384, V, D, 0, 0)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "gelu":
    # FFN-1 forward with the pre-activation (1) or the derivative (6) as its second output; its backward with erf + exp (2) or one multiply (7)
    for tile in ("128", "256"):
        os.environ["TILE"] = tile
        print("TILE =", tile)
        run("fwd ffn1  BIAS_GELU   (u, g)", T, F, D, 0, 0, epi=1)
        run("fwd ffn1  BIAS_GELU_D (g', g)", T, F, D, 0, 0, epi=6)
        run("dX  ffn2->du GELU_BWD (erf)", T, F, D, 0, 1, epi=2)
        run("dX  ffn2->du MUL_AUX  (mul)", T, F, D, 0, 1, epi=7)
        run("dX  plain (no side input)", T, F, D, 0, 1, epi=0)

#!/usr/bin/env python3
"""Variant 2 (256x256 tiles on four waves, csrc/gemm_w4.h) against the default lock-step kernel: results on the forward layouts / epilogues, then
cold-operand timing on the step's k-contiguous shapes."""
import ctypes as C, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gemm_pp_check as G
L = G.L


def case(M, N, K, **kw):
    saved = G.run_gemm
    def rg(variant, *a, **k):
        return saved(2 if variant == 1 else 0, *a, **k)
    G.run_gemm = rg
    try:
        G.case(M, N, K, 0, 0, **kw)
    finally:
        G.run_gemm = saved


if len(sys.argv) == 1:
    for K in (64, 128, 320, 768):
        case(600, 768, K, tag="w4 fwd")
    case(600, 768, 320, resid=True, p_drop=0.1, tag="w4 resid+drop")
    case(600, 776, 192, out_f32=1, tag="w4 f32")
    case(509, 776, 128, epi=1, tag="w4 gelu")
    case(509, 776, 128, epi=2, bias=False, tag="w4 gelu'")
    for M in (2048, 1800, 1500, 1200, 1000):
        case(M, 1024, 256, cu_cap=3, tag="w4 persist")
    case(3000, 1536, 64, cu_cap=5, tag="w4 persist nk=1")
    print("W4 OK")
bf = torch.bfloat16
GP = G.GP


def t(name, M, N, K, epi=0, resid=False, p_drop=0.0, iters=20):
    per = (M * K + N * K + M * N * (2 if epi == 1 else 1)) * 2
    nset = max(2, min(24, int(1.5e9 // per)))
    sets = []
    for _ in range(nset):
        A = torch.randn(M, K, device="cuda").to(bf); B = torch.randn(N, K, device="cuda").to(bf); Cc = torch.empty(M, N, device="cuda", dtype=bf)
        aux = torch.empty(M, N, device="cuda", dtype=bf) if epi == 1 else None
        bias = torch.randn(N, device="cuda"); R = torch.randn(M, N, device="cuda").to(bf) if resid else None
        g = GP(A=A.data_ptr(), B=B.data_ptr(), C=Cc.data_ptr(), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, tile=256, bias=bias.data_ptr(), aux=aux.data_ptr() if aux is not None else 0,
               ldaux=N, R=R.data_ptr() if resid else 0, ldr=N, p_drop=p_drop, seed=7)
        sets.append((g, A, B, Cc, aux, bias, R))
    st = torch.cuda.current_stream().cuda_stream
    res = []
    for variant in (0, 2, 0, 2):
        L.dic_gemm_set_variant(variant)
        for g, *_ in sets: assert L.dic_gemm(1, 0, 0, epi, C.byref(g), st) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = max(iters, nset); e0.record()
        for i in range(n): L.dic_gemm(1, 0, 0, epi, C.byref(sets[i % nset][0]), st)
        e1.record(); torch.cuda.synchronize()
        res.append(2.0 * M * N * K / (e0.elapsed_time(e1) / n) / 1e9)
    L.dic_gemm_set_variant(0)
    print(f"{name:30s} M={M:6d} N={N:6d} K={K:6d}   8-wave {res[0]:7.1f} {res[2]:7.1f}   4-wave {res[1]:7.1f} {res[3]:7.1f} TFLOP/s", flush=True)


T, D, F = 17408, 768, 3072
t("fwd qkv bias", T, 3 * D, D)
t("fwd out-proj +resid+drop", T, D, D, resid=True, p_drop=0.1)
t("fwd ffn1 gelu", T, F, D, epi=1)
t("fwd ffn2 +resid+drop", T, D, F, resid=True, p_drop=0.1)
t("sampling qkv (M=34816)", 34816, 3 * D, D)
t("square 4096", 4096, 4096, 4096)
t("square 8192", 8192, 8192, 8192, iters=5)

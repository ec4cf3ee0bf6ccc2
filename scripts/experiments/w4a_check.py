#!/usr/bin/env python3
"""The hand-scheduled four-wave GEMM (csrc/gemm_w4a.h, dic_gemm_set_w4a) against the default 8-wave kernel: results on eligible shapes
(max difference in bf16 ulps; the bias is added after the K loop instead of before it, so last-bit differences are expected), then
timing, hot and cold operands.    python scripts/experiments/w4a_check.py [time]"""
import ctypes as C, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dic = importlib.import_module("diffusion-image-captioning_amd")
L = dic.lib()
GP = dic._lib.GemmParams
bf = torch.bfloat16
st = lambda: torch.cuda.current_stream().cuda_stream


def run(M, N, K, A, B, bias, mode, Cc=None):
    Cc = torch.full((M, N), float("nan"), dtype=bf, device="cuda") if Cc is None else Cc
    g = GP(A=A.data_ptr(), B=B.data_ptr(), C=Cc.data_ptr(), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias.data_ptr() if bias is not None else 0, tile=256)
    L.dic_gemm_set_w4a(mode)
    rc = L.dic_gemm(1, 0, 0, 0, C.byref(g), st())
    L.dic_gemm_set_w4a(0)
    assert rc == 0, L.dic_last_error()
    return Cc


def check():
    ok = True
    for (M, N, K, with_bias) in [(256, 256, 256, False), (512, 256, 256, True), (256, 512, 768, True), (1024, 768, 768, True), (17408, 2304, 768, True),
                                 (4352, 3072, 768, False), (2048, 2048, 2048, True), (17408, 768, 3072, True), (8192, 8192, 512, False)]:
        g_ = torch.Generator(device="cuda").manual_seed(M + N + K)
        A = (torch.randn(M, K, device="cuda", generator=g_) * 0.5).to(bf)
        B = (torch.randn(N, K, device="cuda", generator=g_) * 0.05).to(bf)
        bias = torch.randn(N, device="cuda", generator=g_) if with_bias else None
        ref = run(M, N, K, A, B, bias, 0)
        got = run(M, N, K, A, B, bias, 1)
        torch.cuda.synchronize()
        exact = (A[:512].double() @ B.double().t() + (bias.double() if with_bias else 0))
        e_ref = float((ref[:512].double() - exact).abs().max() / exact.abs().max())
        e_got = float((got[:512].double() - exact).abs().max() / exact.abs().max())
        nan = int(torch.isnan(got.float()).sum())
        diff = (got.float() - ref.float()).abs()
        ulp = float((diff / (ref.float().abs() * 2 ** -7 + 1e-30)).max())
        nbad = int((diff > ref.float().abs() * 2 ** -6 + 1e-3).sum())
        good = nan == 0 and nbad == 0 and e_got < 6e-3
        ok &= good
        print(f"M={M:6d} N={N:5d} K={K:5d} bias={int(with_bias)}  vs fp64: 8-wave {e_ref:.2e}  w4a {e_got:.2e}   w4a vs 8-wave: max {ulp:.2f} bf16 ulp, {nbad} beyond 2 ulp, {nan} NaN   {'OK' if good else 'MISMATCH'}", flush=True)
    print("W4A OK" if ok else "W4A FAILED")
    return ok


def bench(M, N, K, cold, with_bias=True, iters=20):
    per = (M * K + N * K + M * N) * 2
    nset = max(2, min(24, int(1.5e9 // per))) if cold else 1
    sets = []
    for _ in range(nset):
        A = torch.randn(M, K, device="cuda").to(bf); B = (torch.randn(N, K, device="cuda") * 0.05).to(bf)
        Cc = torch.empty(M, N, dtype=bf, device="cuda"); bias = torch.randn(N, device="cuda") if with_bias else None
        sets.append((A, B, bias, Cc))
    out = []
    for mode in (0, 1, 0, 1):
        for s_ in sets:
            run(M, N, K, *s_[:3], mode, s_[3])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = max(iters, nset)
        L.dic_gemm_set_w4a(mode)
        gs = [GP(A=s_[0].data_ptr(), B=s_[1].data_ptr(), C=s_[3].data_ptr(), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=s_[2].data_ptr() if s_[2] is not None else 0, tile=256) for s_ in sets]
        e0.record()
        for i in range(n):
            L.dic_gemm(1, 0, 0, 0, C.byref(gs[i % nset]), st())
        e1.record()
        torch.cuda.synchronize()
        L.dic_gemm_set_w4a(0)
        out.append(2.0 * M * N * K / (e0.elapsed_time(e1) / n) / 1e9)
    print(f"M={M:6d} N={N:5d} K={K:5d} {'cold' if cold else 'hot '}  8-wave {out[0]:7.1f} {out[2]:7.1f}   w4a {out[1]:7.1f} {out[3]:7.1f} TFLOP/s   ({2.0 * M * N * K / out[3] / 1e6:.1f} us)", flush=True)


if __name__ == "__main__":
    good = check()
    if len(sys.argv) > 1 and sys.argv[1] == "time" and good:
        for cold in (False, True):
            bench(17408, 2304, 768, cold)
            bench(17408, 768, 768, cold)
            bench(17408, 3072, 768, cold)
            bench(17408, 768, 3072, cold)
            bench(34816, 2304, 768, cold)
            bench(4096, 4096, 4096, cold, False)
            bench(8192, 8192, 8192, cold, False, iters=5)
    sys.exit(0 if good else 1)

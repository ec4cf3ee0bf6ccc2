#!/usr/bin/env python3
"""The hand-scheduled four-wave GEMM (csrc/gemm_w4a.h, dic_gemm_set_w4a) against the default 8-wave kernel: results on eligible shapes for every
variant (B k-contiguous / k-major x plain / + residual / x aux; differences beyond 2 bf16 ulp are errors -- the bias is added after the K loop
instead of before it, so last-bit differences are expected), then timing with hot and cold operands.
    python scripts/experiments/w4a_check.py [time] [rows=224|256]"""
import ctypes as C, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dic = importlib.import_module("diffusion-image-captioning_amd")
L = dic.lib()
GP = dic._lib.GemmParams
ROWS = next((int(x[5:]) for x in sys.argv if x.startswith("rows=")), 0)       # rows=224 | rows=256: force that tile height of the asm kernel (default: per launch)
assert L.dic_set_option(b"gemm_w4a_rows", ROWS) == 0
bf = torch.bfloat16
st = lambda: torch.cuda.current_stream().cuda_stream
EPI_OF = {"plain": 0, "resid": 0, "mulaux": 7, "dropres": 0}


def make(M, N, K, bkm, kind, seed=0):
    g_ = torch.Generator(device="cuda").manual_seed(M + N + K + seed)
    A = (torch.randn(M, K, device="cuda", generator=g_) * 0.5).to(bf)
    B = (torch.randn((K, N) if bkm else (N, K), device="cuda", generator=g_) * 0.05).to(bf)
    bias = torch.randn(N, device="cuda", generator=g_) if kind != "mulaux" and (seed % 2 == 0) else None
    side = torch.randn(M, N, device="cuda", generator=g_).to(bf) if kind != "plain" else None
    return A, B, bias, side


def params(M, N, K, bkm, kind, A, B, bias, side, Cc):
    g = GP(A=A.data_ptr(), B=B.data_ptr(), C=Cc.data_ptr(), M=M, N=N, K=K, lda=K, ldb=(N if bkm else K), ldc=N, bias=bias.data_ptr() if bias is not None else 0, tile=256)
    if kind in ("resid", "dropres"):
        g.R, g.ldr = side.data_ptr(), N
    if kind == "dropres":
        g.p_drop, g.seed = 0.1, 0x1234567890 + M
    if kind == "mulaux":
        g.aux, g.ldaux = side.data_ptr(), N
    return g


def run(M, N, K, bkm, kind, ops, mode, Cc=None):
    Cc = torch.full((M, N), float("nan"), dtype=bf, device="cuda") if Cc is None else Cc
    g = params(M, N, K, bkm, kind, *ops, Cc)
    L.dic_gemm_set_w4a(mode)
    rc = L.dic_gemm(1, 0, bkm, EPI_OF[kind], C.byref(g), st())
    L.dic_gemm_set_w4a(0)
    assert rc == 0, L.dic_last_error()
    return Cc


def check():
    ok = True
    shapes = [(256, 256, 256), (512, 256, 256), (256, 512, 768), (1024, 768, 768), (17408, 2304, 768), (4352, 3072, 768), (2048, 2048, 2048), (17408, 768, 3072), (2304, 768, 2304)]
    for bkm in (0, 1):
        for kind in ("plain", "resid", "mulaux") + (() if bkm else ("dropres",)):
            for n_, (M, N, K) in enumerate(shapes):
                ops = make(M, N, K, bkm, kind, n_)
                ref = run(M, N, K, bkm, kind, ops, 0)
                got = run(M, N, K, bkm, kind, ops, 1)
                torch.cuda.synchronize()
                nan = int(torch.isnan(got.float()).sum())
                diff = (got.float() - ref.float()).abs()
                nbad = int((diff > ref.float().abs() * 2 ** -6 + 2e-3).sum())
                good = nan == 0 and nbad == 0
                ok &= good
                print(f"B {'KM' if bkm else 'KC'} {kind:7s} M={M:6d} N={N:5d} K={K:5d} bias={int(ops[2] is not None)}: max |diff| {float(diff.max()):.3e}, {nbad} beyond 2 ulp, {nan} NaN   {'OK' if good else 'MISMATCH'}", flush=True)
    print("W4A OK" if ok else "W4A FAILED")
    return ok


def bench(M, N, K, bkm, kind, cold, iters=20):
    per = (M * K + N * K + M * N * (2 if kind != "plain" else 1)) * 2
    nset = max(2, min(24, int(1.5e9 // per))) if cold else 1
    sets = [(make(M, N, K, bkm, kind, s_), torch.empty(M, N, dtype=bf, device="cuda")) for s_ in range(nset)]
    out = []
    for mode in (0, 1, 0, 1):
        for ops, Cc in sets:
            run(M, N, K, bkm, kind, ops, mode, Cc)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = max(iters, nset)
        L.dic_gemm_set_w4a(mode)
        gs = [params(M, N, K, bkm, kind, *ops, Cc) for ops, Cc in sets]
        e0.record()
        for i in range(n):
            L.dic_gemm(1, 0, bkm, EPI_OF[kind], C.byref(gs[i % nset]), st())
        e1.record()
        torch.cuda.synchronize()
        L.dic_gemm_set_w4a(0)
        out.append(2.0 * M * N * K / (e0.elapsed_time(e1) / n) / 1e9)
    print(f"B {'KM' if bkm else 'KC'} {kind:7s} M={M:6d} N={N:5d} K={K:5d} {'cold' if cold else 'hot '}  8-wave {out[0]:7.1f} {out[2]:7.1f}   w4a {out[1]:7.1f} {out[3]:7.1f} TFLOP/s   ({2.0 * M * N * K / out[3] / 1e6:.1f} us)", flush=True)


if __name__ == "__main__":
    good = check()
    if len(sys.argv) > 1 and sys.argv[1] == "time" and good:
        for cold in (False, True):
            bench(17408, 2304, 768, 0, "plain", cold)           # QKV forward
            bench(17408, 768, 768, 1, "plain", cold)            # out-proj dX
            bench(17408, 768, 3072, 1, "resid", cold)           # FFN-1 dX + residual
            bench(17408, 768, 2304, 1, "resid", cold)           # QKV dX + residual
            bench(17408, 3072, 768, 1, "mulaux", cold)          # FFN-2 dX x gelu'
            bench(17408, 768, 3072, 0, "resid", cold)           # FFN-2 forward without dropout (eval / sampling)
            bench(17408, 768, 3072, 0, "dropres", cold)         # FFN-2 forward as trained (dropout 0.1 + residual)
            bench(17408, 768, 768, 0, "dropres", cold)          # out-proj forward as trained
            bench(4096, 4096, 4096, 0, "plain", cold)
            bench(8192, 8192, 8192, 0, "plain", cold, iters=5)
    sys.exit(0 if good else 1)

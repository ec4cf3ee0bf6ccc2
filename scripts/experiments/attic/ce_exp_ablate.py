#!/usr/bin/env python3
"""Times the rounding head's forward GEMM (M=16384, V=30522, K=768, tile 256) with the CE_EXP epilogue of every ab/ce_*/libdic_hip.so, next to
CE_DLOGITS (exp + 1 GB store) and the plain bf16 store of the same tile on the shipped library.  One process per library (DIC_HIP_LIB)."""
import ctypes as C, glob, importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) == 1:
    for lib in [""] + sorted(glob.glob(os.path.join(ROOT, "ab", "ce_*", "libdic_hip.so"))):
        env = dict(os.environ)
        if lib:
            env["DIC_HIP_LIB"] = lib
        for _ in range(1):
            subprocess.run([sys.executable, os.path.abspath(__file__), lib or "shipped"], env=env)
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
dic = importlib.import_module("diffusion-image-captioning_amd")
L = dic.lib()
GP = dic._lib.GemmParams
M, V, K = 16384, 30522, 768
vpad = (V + 127) // 128 * 128
bf = torch.bfloat16
x = torch.randn(M, K, device="cuda").to(bf); W = torch.zeros(vpad, K, device="cuda", dtype=bf); W[:V] = (torch.randn(V, K, device="cuda") * 0.05).to(bf)
E = torch.empty(M, vpad, device="cuda", dtype=bf)
tgt = torch.randint(0, V, (M,), device="cuda"); c = torch.full((M,), 40.0, device="cuda"); lse = torch.full((M,), 11.0, device="cuda")
part = torch.empty(M * 480 * 4, device="cuda"); tl = torch.zeros(M, device="cuda")
st = torch.cuda.current_stream().cuda_stream


def t(epi, **kw):
    g = GP(A=x.data_ptr(), B=W.data_ptr(), C=E.data_ptr(), M=M, N=V, K=K, lda=K, ldb=K, ldc=vpad, tile=256, **kw)
    for _ in range(3):
        assert L.dic_gemm(1, 0, 0, epi, C.byref(g), st) == 0, L.dic_last_error()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        L.dic_gemm(1, 0, 0, epi, C.byref(g), st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3


def t0():
    g = GP(A=x.data_ptr(), B=W.data_ptr(), C=E.data_ptr(), M=M, N=vpad, K=K, lda=K, ldb=K, ldc=vpad, tile=256)
    for _ in range(3): L.dic_gemm(1, 0, 0, 0, C.byref(g), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(10): L.dic_gemm(1, 0, 0, 0, C.byref(g), st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3


t5 = lambda: t(5, tgt=tgt.data_ptr(), lse=c.data_ptr(), partial=part.data_ptr(), tgt_logit=tl.data_ptr())
t4 = lambda: t(4, tgt=tgt.data_ptr(), lse=lse.data_ptr(), ce_rows_a=M, ce_scale_a=1.0, ce_scale_b=1.0)
for _ in range(3):          # clocks up before anything is recorded
    t0()
r5, r4, r0 = [], [], []
for _ in range(3):          # interleaved, best of three
    r4.append(t4()); r5.append(t5()); r0.append(t0())
name = "shipped" if sys.argv[1] == "shipped" else os.path.basename(os.path.dirname(sys.argv[1]))
print(f"{name:24s} CE_EXP {min(r5):7.1f} us   CE_DLOGITS {min(r4):7.1f} us   plain bf16 store {min(r0):7.1f} us", flush=True)

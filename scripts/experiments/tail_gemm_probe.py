#!/usr/bin/env python3
"""The CLIP-projection weight gradient at the end of the backward (fp32, k-major A with a 17 x 768-float row stride): time per split factor."""
import ctypes as C, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dic = importlib.import_module("diffusion-image-captioning_amd")
L = dic.lib(); GP = dic._lib.GemmParams
N, D, Tk = 1024, 768, 17
dy0 = torch.randn(N, Tk * D, device="cuda"); img = torch.randn(N, 512, device="cuda"); G = torch.empty(D, 512, device="cuda")
ws = torch.empty(64 * D * 512, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for lda, name in ((Tk * D, "strided rows (as in the step)"), (D, "dense rows")):
    for sk in (1, 2, 4, 8, 16, 32):
        g = GP(A=dy0.data_ptr(), B=img.data_ptr(), C=G.data_ptr(), M=D, N=512, K=N, lda=lda, ldb=512, ldc=512, out_f32=1, split_k=sk, split_ws=ws.data_ptr() if sk > 1 else 0)
        for _ in range(5): assert L.dic_gemm(0, 1, 1, 0, C.byref(g), st) == 0, L.dic_last_error()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(50): L.dic_gemm(0, 1, 1, 0, C.byref(g), st)
        e1.record(); torch.cuda.synchronize()
        print(f"{name:30s} split {sk:2d}: {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us (GEMM + fold)", flush=True)

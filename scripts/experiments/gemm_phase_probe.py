#!/usr/bin/env python3
"""Which side of the K-loop bounds the LDS-DMA GEMM?  Needs the debug build (ab/libdic_dbg.so; cu_cap bits 16/17 switch the DMA
issue / the LDS-read+MFMA body off).  Results are garbage numerically; only the timing matters."""
import ctypes as C, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dic = importlib.import_module("diffusion-image-captioning_amd")
L = dic.lib(); GP = dic._lib.GemmParams
bf = torch.bfloat16
def run(M, N, K, tile, dbg, a_km=0, b_km=0, iters=20):
    A = torch.randn((K, M) if a_km else (M, K), device="cuda").to(bf); B = torch.randn((K, N) if b_km else (N, K), device="cuda").to(bf)
    Cc = torch.empty(M, N, device="cuda", dtype=bf)
    g = GP(A=A.data_ptr(), B=B.data_ptr(), C=Cc.data_ptr(), M=M, N=N, K=K, lda=A.shape[1], ldb=B.shape[1], ldc=N, tile=tile, cu_cap=dbg << 16)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3): assert L.dic_gemm(1, a_km, b_km, 0, C.byref(g), st) == 0, L.dic_last_error()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): L.dic_gemm(1, a_km, b_km, 0, C.byref(g), st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
names = {0: "full", 11: "epilogue only", 11 + 32: "epi w/o LDS", 11 + 32 + 64: "epi w/o LDS, tile-contiguous stores", 11 + 64: "epilogue only, tile-contig", 64: "full, tile-contig", 15: "nothing"}
for (M, N, K) in ((18432, 768, 768), (18432, 3072, 768)):
    for tile in (256, 128):
        res = {d: run(M, N, K, tile, d) for d in names}
        print(f"M={M} N={N} K={K} tile={tile}: " + "  ".join(f"{names[d]} {res[d]:6.1f} us" for d in res), flush=True)

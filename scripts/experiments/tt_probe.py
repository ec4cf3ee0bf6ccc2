#!/usr/bin/env python3
"""K-loop cost per layout (needs ab/libdic_dbg.so: cu_cap bit 16 = no DMA issue, bit 17 = no LDS-read/MFMA body)."""
import ctypes as C, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dic = importlib.import_module("diffusion-image-captioning_amd")
L = dic.lib(); GP = dic._lib.GemmParams
bf = torch.bfloat16
def run(M, N, K, tile, dbg, a_km, b_km, iters=10):
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
M = N = 4096; K = 8192; nk = K // 64
for tile in (256, 128):
    rounds = (M // tile) * (N // tile) / (256 if tile == 256 else 512)
    for name, (a, b) in {"NN (KC,KC)": (0, 0), "NT (KC,KM)": (0, 1), "TT (KM,KM)": (1, 1)}.items():
        r = {d: run(M, N, K, tile, d, a, b) for d in (0, 1, 2, 3)}
        print(f"tile {tile} {name}: full {r[0]:7.1f} us | per K-step per round: full {(r[0]-r[3])/nk/rounds:5.2f}  LDS+MFMA only {(r[1]-r[3])/nk/rounds:5.2f}  DMA only {(r[2]-r[3])/nk/rounds:5.2f} us", flush=True)

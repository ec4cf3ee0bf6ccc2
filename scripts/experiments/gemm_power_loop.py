#!/usr/bin/env python3
"""One GEMM shape in a loop for a few seconds (8-wave kernel or the four-wave asm kernel), printing TFLOP/s per second of wall time -- to be sampled by
rocm-smi from outside (scripts/experiments/power_probe.sh): what do the two kernels sustain at the package power limit, and at which clock?
    python scripts/experiments/gemm_power_loop.py <w4a 0|1> [M N K] [seconds]"""
import ctypes as C, importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dic = importlib.import_module("diffusion-image-captioning_amd")
L = dic.lib()
GP = dic._lib.GemmParams
mode = int(sys.argv[1])
M, N, K = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (8192, 8192, 8192)
secs = float(sys.argv[5]) if len(sys.argv) > 5 else 8.0
bf = torch.bfloat16
sets = []
for _ in range(3):
    A = torch.randn(M, K, device="cuda").to(bf); B = (torch.randn(N, K, device="cuda") * 0.05).to(bf); Cc = torch.empty(M, N, dtype=bf, device="cuda")
    sets.append((A, B, Cc, GP(A=A.data_ptr(), B=B.data_ptr(), C=Cc.data_ptr(), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, tile=256)))
st = torch.cuda.current_stream().cuda_stream
L.dic_gemm_set_w4a(mode)
t_end = time.time() + secs
i = 0
while time.time() < t_end:
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 1.0:
        for _ in range(20):
            L.dic_gemm(1, 0, 0, 0, C.byref(sets[i % 3][3]), st); i += 1; n += 1
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{'w4a' if mode else '8-wave'} M={M} N={N} K={K}: {2.0 * M * N * K * n / dt / 1e12:7.1f} TFLOP/s sustained", flush=True)

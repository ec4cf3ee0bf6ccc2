#!/usr/bin/env python3
"""What the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS, as shipped in this image) reaches on the step's shapes, cold operands -- a measured
reference for what this chip's GEMM path gives on THESE shapes (cdna_hip_programming.md rule 10: never infer a ceiling from one's own kernels).
Measurement only: nothing in the product calls a vendor library."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
bf = torch.bfloat16
T, D, F, V = 17408, 768, 3072, 30592


def run(name, M, N, K, ta=False, tb=True, out_dtype=bf, iters=20):
    per = (M * K + N * K + M * N) * 2
    nset = max(2, min(24, int(1.5e9 // per)))
    sets = []
    for _ in range(nset):
        A = torch.randn((K, M) if ta else (M, K), device="cuda").to(bf)
        B = torch.randn((N, K) if tb else (K, N), device="cuda").to(bf)
        sets.append((A.t() if ta else A, B.t() if tb else B))
    for a, b in sets:
        torch.matmul(a, b)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = max(iters, nset)
        e0.record()
        for i in range(n):
            torch.matmul(*sets[i % nset])
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    print(f"{name:34s} M={M:6d} N={N:6d} K={K:6d}  {best*1e3:8.1f} us  {2.0*M*N*K/best/1e9:7.1f} TFLOP/s   (torch.matmul, no epilogue, cold, {nset} sets)", flush=True)


print("torch", torch.__version__, "| preferred blas:", torch.backends.cuda.preferred_blas_library() if hasattr(torch.backends.cuda, "preferred_blas_library") else "?")
run("fwd qkv        x[M,K] W[N,K]^T", T, 3 * D, D)
run("fwd out-proj", T, D, D)
run("fwd ffn1", T, F, D)
run("fwd ffn2", T, D, F)
run("dX  ffn2->du   dY[M,K] W[K,N]", T, F, D, tb=False)
run("dX  ffn1->dsa", T, D, F, tb=False)
run("dX  qkv->dh", T, D, 3 * D, tb=False)
run("dW  ffn1       dY[K,M]^T X[K,N]", F, D, T, ta=True, tb=False)
run("dW  qkv", 3 * D, D, T, ta=True, tb=False)
run("dW  out-proj", D, D, T, ta=True, tb=False)
run("rounding logits", 16384, V, D, iters=6)
run("rounding dX", 16384, D, V, tb=False, iters=6)
run("square 4096", 4096, 4096, 4096)
run("square 8192", 8192, 8192, 8192, iters=5)

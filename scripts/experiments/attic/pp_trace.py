#!/usr/bin/env python3
"""Reads the s_memtime trace of the DIC_PP_TRACE build (ab/pp_trace/libdic_hip.so; run with DIC_HIP_LIB pointing at it): per segment of the
ping-pong K loop, the cycles wave 0 / wave 4 of workgroup 0 spend working (barrier exit -> next barrier arrival) and waiting at the barrier."""
import ctypes as C, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dic = importlib.import_module("diffusion-image-captioning_amd")
L = dic.lib(); GP = dic._lib.GemmParams; bf = torch.bfloat16
L.dic_gemm_set_variant(1)
for (M, N, K) in [(8192, 8192, 8192), (17408, 2304, 768)]:
    A = torch.randn(M, K, device="cuda").to(bf); B = torch.randn(N, K, device="cuda").to(bf); Cc = torch.empty(M, N, device="cuda", dtype=bf)
    tr = torch.zeros(512, dtype=torch.int64, device="cuda")
    g = GP(A=A.data_ptr(), B=B.data_ptr(), C=Cc.data_ptr(), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, tile=256, tgt_logit=tr.data_ptr())
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        tr.zero_()
        assert L.dic_gemm(1, 0, 0, 0, C.byref(g), st) == 0
        torch.cuda.synchronize()
    t = tr.cpu().numpy().reshape(2, 256)
    print(f"== M={M} N={N} K={K}: stamps per K-step = 16 pairs (before barrier, after barrier); segments L0 M0 L1 M1 L2 M2 L3 M3")
    names = ["L0", "M0", "L1", "M1", "L2", "M2", "L3", "M3"]
    for w in range(2):
        s = t[w]
        n = int((s != 0).sum())
        ks = n // 16
        print(f" wave {4*w}: {n} stamps, K-steps {ks}")
        for k in range(1, ks):
            row = s[16 * k: 16 * k + 16]
            prev_after = s[16 * k - 1]
            out = []
            for seg in range(8):
                before, after = row[2 * seg], row[2 * seg + 1]
                work = before - prev_after
                wait = after - before
                out.append(f"{names[seg]} {work:4d}+{wait:4d}")
                prev_after = after
            print(f"   k{k}: " + " | ".join(out) + f"   total {row[15] - s[16 * k - 1]}")

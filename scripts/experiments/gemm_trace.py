#!/usr/bin/env python3
"""Per-phase timeline of the bf16 GEMM from s_memtime stamps (needs the trace build: csrc/gemm.hip compiled with -DDIC_GEMM_TRACE,
linked into ab/libdic_trace.so and loaded with DIC_HIP_LIB).  Wave 0 of every workgroup stamps: kernel entry; per tile: first K-step
landed (loop top) / K loop done / epilogue issued.  Prints the mean duration of each phase over workgroups, in s_memtime ticks (shader clock; the build also stamps s_memrealtime, 100 MHz, at entry and exit so the
clock the kernel actually ran at is printed) and the spread of the loop-top stamp across workgroups (how much in lockstep the chip runs)."""
import ctypes as C, importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dic = importlib.import_module("diffusion-image-captioning_amd")
L = dic.lib(); GP = dic._lib.GemmParams
bf = torch.bfloat16


def run(name, M, N, K, a_km=0, b_km=0, epi=0, resid=False, tile=256):
    A = torch.randn((K, M) if a_km else (M, K), device="cuda").to(bf); B = torch.randn((K, N) if b_km else (N, K), device="cuda").to(bf)
    Cc = torch.empty(M, N, device="cuda", dtype=bf)
    aux = torch.randn(M, N, device="cuda").to(bf) if epi in (1, 2) else None
    bias = torch.randn(N, device="cuda") if epi in (0, 1) else None
    Rr = torch.randn(M, N, device="cuda").to(bf) if resid else None
    tr = torch.zeros(2048 * 64, dtype=torch.int64, device="cuda")
    ks = torch.zeros(2048 * 2 * 16 * 4, dtype=torch.int64, device="cuda")
    g = GP(A=A.data_ptr(), B=B.data_ptr(), C=Cc.data_ptr(), M=M, N=N, K=K, lda=A.shape[1], ldb=B.shape[1], ldc=N, tile=tile,
           bias=bias.data_ptr() if bias is not None else 0, aux=aux.data_ptr() if aux is not None else 0, ldaux=N,
           R=Rr.data_ptr() if resid else 0, ldr=N, tgt_logit=tr.data_ptr(), partial=ks.data_ptr() if epi in (0, 1, 2) else 0, cu_cap=int(os.environ.get("CU_CAP", "0")))
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        assert L.dic_gemm(1, a_km, b_km, epi, C.byref(g), st) == 0, L.dic_last_error()
    torch.cuda.synchronize()
    tr.zero_(); ks.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); L.dic_gemm(1, a_km, b_km, epi, C.byref(g), st); e1.record()
    torch.cuda.synchronize()
    t = tr.cpu().numpy().reshape(2048, 64)
    wg = t[t[:, 0] > 0]
    base = wg[:, 0].min()
    clk = (wg[:, 61] - wg[:, 0]) / np.maximum(wg[:, 63] - wg[:, 62], 1) * 0.1      # shader ticks per 10 ns tick of the 100 MHz clock -> GHz
    wg = wg.copy(); wg[:, 61:] = 0
    ntile = ((wg != 0).sum(1) - 1) // 6
    print(f"== {name}: M={M} N={N} K={K}  {len(wg)} workgroups, tiles/workgroup {ntile.min()}..{ntile.max()}, event time {e0.elapsed_time(e1)*1e3:.1f} us")
    print(f"   shader clock while the kernel ran: {np.median(clk):.3f} GHz (min {clk.min():.3f}, max {clk.max():.3f})")
    # stamps per tile (wave 0): own loads+stores drained | barrier passed (loop top) | K loop done | epilogue: side loads in | next tile's DMA issued | stores issued
    for k in range(int(ntile.max())):
        sel = wg[ntile > k]
        prev, vm, top, kend, eb, ei, eend = (sel[:, j + 6 * k] for j in range(7))
        print(f"   tile {k}: drain {np.mean(vm-prev)/1000:5.2f} + barrier {np.mean(top-vm)/1000:5.2f} | K loop {np.mean(kend-top)/1000:6.2f} | epilogue: to loads-in {np.mean(eb-kend)/1000:5.2f}"
              f" + next-tile set-up/DMA issue {np.mean(ei-eb)/1000:5.2f} + math/stores {np.mean(eend-ei)/1000:5.2f} kcyc")

    # K-step anatomy of the first tile (waves 0 and NW-1 of every workgroup): compute | wait for the next stage's DMA | wait at the barrier
    k = ks.cpu().numpy().reshape(2048, 2, 16, 4)[: len(wg)]
    for w in (0, 1):
        kk = k[:, w]
        nst = int((kk[0, :, 0] > 0).sum())
        if nst < 3:
            continue
        comp = kk[:, 1:nst, 0] - kk[:, 0:nst - 1, 2]           # barrier release of step s-1 -> this wave done computing step s
        vm = kk[:, :nst, 1] - kk[:, :nst, 0]
        bar = kk[:, :nst, 2] - kk[:, :nst, 1]
        print(f"   K-steps of tile 0, wave {'0' if w == 0 else 'last'}: compute {comp.mean():7.0f} cyc (min {comp.min()}, max {comp.max()}) | vmcnt wait {vm[:, 1:].mean():6.0f}"
              f" (step 0: {vm[:, 0].mean():6.0f}) | barrier wait {bar[:, 1:].mean():6.0f} | per step {np.mean(kk[:, 1:nst, 2] - kk[:, 0:nst - 1, 2]):7.0f}")


T = int(os.environ.get("TOKENS", "17408"))
run("fwd qkv", T, 2304, 768)
run("fwd out-proj +resid", T, 768, 768, resid=True)
run("fwd ffn1 gelu", T, 3072, 768, epi=1)
run("fwd ffn2 +resid", T, 768, 3072, resid=True)
run("dX gelu'", T, 3072, 768, b_km=1, epi=2)
run("dX ffn1", T, 768, 3072, b_km=1, resid=True)
run("square 4096", 4096, 4096, 4096)

#!/usr/bin/env python3
"""Standalone time of the bf16 LayerNorm backward (17 408 x 768, 108 MB of traffic) against the number of persistent blocks, for the shipped kernel
(DIC_LN_BWD_ROWS unset) and the thin form (DIC_LN_BWD_ROWS=2 / 4: rows per wave iteration): can the kernel saturate HBM from a fraction of the CUs?
    scripts/build_variant.sh thin "-DDIC_LN_THIN"; for r in 1 2 4; do DIC_HIP_LIB=abl/libdic_thin.so DIC_LN_BWD_ROWS=$r python scripts/experiments/ln_bwd_thin_probe.py; done"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dic = importlib.import_module("diffusion-image-captioning_amd")
L = dic.lib()
T, D = int(os.environ.get("T", "17408")), 768
g = torch.Generator().manual_seed(0)
y = (torch.randn(T, D, generator=g) * 1.3).to("cuda", torch.bfloat16)
dh = torch.randn(T, D, generator=g).to("cuda", torch.bfloat16)
gamma = (1 + 0.1 * torch.randn(D, generator=g)).cuda()
mean, rstd = torch.zeros(T, device="cuda"), torch.zeros(T, device="cuda")
h = torch.zeros(T, D, dtype=torch.bfloat16, device="cuda")
st = torch.cuda.current_stream().cuda_stream
assert L.dic_ln_fwd(1, y.data_ptr(), gamma.data_ptr(), gamma.data_ptr(), h.data_ptr(), mean.data_ptr(), rstd.data_ptr(), T, D, 1e-12, st) == 0
dx = torch.zeros(T, D, dtype=torch.bfloat16, device="cuda")
ref = None
rows = os.environ.get("DIC_LN_BWD_ROWS", "1")
for npart in (32, 64, 128, 256, 512, 1024):
    part = torch.zeros(npart, 3 * D, device="cuda")
    call = lambda: L.dic_ln_bwd(1, dh.data_ptr(), y.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), 0, 0.0, 0,
                                part.data_ptr(), npart, T, D, st)
    for _ in range(5):
        assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10
    chk = (float(dx.float().abs().sum()), float(part.sum()))
    if ref is None:
        ref = chk
    print(f"rows/iter {rows}  blocks {npart:5d}: {us:7.1f} us  {3 * T * D * 2 / us / 1e6:5.2f} TB/s   dx |sum| {chk[0]:.6e}  partial sum {chk[1]:.6e}", flush=True)

#!/usr/bin/env python3
"""Time dic_lin_prep alone on the step's four shapes (us per launch, back to back on one stream), optionally behind a kernel that has just
written the input from all CUs (--cold: what the step sees).   DIC_HIP_LIB=<variant> python scripts/experiments/lin_prep_probe.py"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dic = importlib.import_module("diffusion-image-captioning_amd")
L = dic.lib()
T = 17408
st = torch.cuda.current_stream().cuda_stream
err = torch.zeros(1, dtype=torch.int32).pin_memory()
line = os.environ.get("DIC_HIP_LIB", "shipped").split("/")[-1] + ":"
for K, N, resid in ((768, 2304, 0), (768, 768, 1), (768, 3072, 0), (3072, 768, 1)):
    A = torch.randn(T, K, device="cuda").to(torch.bfloat16)
    hi, lo = (torch.randn(N, K, device="cuda").to(torch.bfloat16) for _ in range(2))
    bias, rref, bin_, bpost, yref = (torch.zeros(N, device="cuda") for _ in range(5))
    ws = torch.zeros(L.dic_lin_prep_ws_bytes(K) // 4, device="cuda")
    def call():
        rc = L.dic_lin_prep(A.data_ptr(), T, K, 16, K, hi.data_ptr() if resid else 0, lo.data_ptr(), K, N, bias.data_ptr(), rref.data_ptr() if resid else 0, 1 if resid else 0,
                            bin_.data_ptr(), 0, yref.data_ptr() if resid else 0, ws.data_ptr(), st)
        assert rc == 0, L.dic_last_error()
    for cold in (0, 1):
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 100
        if cold:            # a streaming kernel rewrites A between the launches (its time is measured separately and subtracted)
            e0.record()
            for _ in range(n):
                A.add_(0)
            e1.record(); torch.cuda.synchronize()
            base = e0.elapsed_time(e1)
            e0.record()
            for _ in range(n):
                A.add_(0); call()
            e1.record(); torch.cuda.synchronize()
            us = (e0.elapsed_time(e1) - base) * 1e3 / n
        else:
            e0.record()
            for _ in range(n):
                call()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
        line += f"  K{K} N{N}{' cold' if cold else ''} {us:5.1f}"
    ws.zero_()
print(line, "us", flush=True)

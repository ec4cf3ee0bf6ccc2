#!/usr/bin/env python3
"""Streaming-write / copy rates at the sizes of the GEMM outputs (what an epilogue could reach at best)."""
import torch
def t(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for mb in (28, 85, 113, 226, 1024):
    n = mb * 1024 * 1024 // 2
    a = torch.empty(n, dtype=torch.bfloat16, device="cuda"); b = torch.empty_like(a)
    tf = t(lambda: a.fill_(1.0)); tc = t(lambda: b.copy_(a))
    print(f"{mb:5d} MB: fill {tf:7.1f} us = {mb*1.048576/tf*1e3/1e3:6.2f} TB/s   copy {tc:7.1f} us = {2*mb*1.048576/tc*1e3/1e3:6.2f} TB/s (r+w)")

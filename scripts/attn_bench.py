#!/usr/bin/env python3
"""Attention microbenchmark at the step's shape (1024 sequences x 18 tokens x 12 heads): fwd/bwd, with and without dropout."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dic = importlib.import_module("diffusion-image-captioning_amd")
L = dic.lib()
N, Tk, H = int(os.environ.get("N", 1024)), int(os.environ.get("TK", 18)), 12
bf = torch.bfloat16
qkv = torch.randn(N * Tk, 3 * H * 64, device="cuda").to(bf); dctx = torch.randn(N * Tk, H * 64, device="cuda").to(bf)
ctx = torch.empty_like(dctx); dqkv = torch.empty_like(qkv)
mask = torch.ones(N, Tk, dtype=torch.uint8, device="cuda"); mask[:, Tk - 1] = 0
st = torch.cuda.current_stream().cuda_stream
def t(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
mb_f = (qkv.numel() + ctx.numel()) * 2 / 1e6; mb_b = (2 * qkv.numel() + dctx.numel()) * 2 / 1e6
for p in (0.0, 0.1):
    f = t(lambda: L.dic_attn_fwd(1, qkv.data_ptr(), mask.data_ptr(), ctx.data_ptr(), N, Tk, H, 64, p, 5, st))
    b = t(lambda: L.dic_attn_bwd(1, qkv.data_ptr(), mask.data_ptr(), dctx.data_ptr(), dqkv.data_ptr(), N, Tk, H, 64, p, 5, st))
    print(f"p_drop={p}: fwd {f:6.1f} us ({mb_f/f:5.2f} TB/s of {mb_f:.0f} MB)   bwd {b:6.1f} us ({mb_b/b:5.2f} TB/s of {mb_b:.0f} MB)")

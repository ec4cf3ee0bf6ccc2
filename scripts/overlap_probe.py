#!/usr/bin/env python3
"""Does splitting the batch in two and interleaving the halves on two streams (each capped to half the CUs) beat one full-width
launch sequence?  Probe with the forward chain of one layer: qkv GEMM -> (LN-like streaming kernel) -> out-proj -> ffn1(gelu) -> ffn2."""
import ctypes as C, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dic = importlib.import_module("diffusion-image-captioning_amd")
L = dic.lib(); GP = dic._lib.GemmParams
D, F = 768, 3072
bf = torch.bfloat16

def mk(T):
    d = dict(h=torch.randn(T, D, device="cuda").to(bf), qkv=torch.empty(T, 3*D, device="cuda", dtype=bf), ctx=torch.randn(T, D, device="cuda").to(bf),
             y=torch.empty(T, D, device="cuda", dtype=bf), sa=torch.empty(T, D, device="cuda", dtype=bf), u=torch.empty(T, F, device="cuda", dtype=bf),
             g=torch.empty(T, F, device="cuda", dtype=bf), y2=torch.empty(T, D, device="cuda", dtype=bf), mean=torch.empty(T, device="cuda"), rstd=torch.empty(T, device="cuda"))
    return d
W = dict(qkv=torch.randn(3*D, D, device="cuda").to(bf), o=torch.randn(D, D, device="cuda").to(bf), w1=torch.randn(F, D, device="cuda").to(bf), w2=torch.randn(D, F, device="cuda").to(bf))
bias = torch.zeros(F, device="cuda"); gam = torch.ones(D, device="cuda")

def gemm(A, B, Cc, M, N, K, st, epi=0, aux=None, cap=0, tile=256):
    g = GP(A=A.data_ptr(), B=B.data_ptr(), C=Cc.data_ptr(), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias.data_ptr(), aux=aux.data_ptr() if aux is not None else 0, ldaux=N, tile=tile, cu_cap=cap)
    assert L.dic_gemm(1, 0, 0, epi, C.byref(g), st) == 0

def chain(d, T, stream, cap):
    st = stream.cuda_stream
    gemm(d["h"], W["qkv"], d["qkv"], T, 3*D, D, st, cap=cap)
    gemm(d["ctx"], W["o"], d["y"], T, D, D, st, cap=cap)
    L.dic_ln_fwd(1, d["y"].data_ptr(), gam.data_ptr(), gam.data_ptr(), d["sa"].data_ptr(), d["mean"].data_ptr(), d["rstd"].data_ptr(), T, D, C.c_float(1e-12), st)
    gemm(d["sa"], W["w1"], d["g"], T, F, D, st, epi=1, aux=d["u"], cap=cap, tile=128)
    gemm(d["g"], W["w2"], d["y2"], T, D, F, st, cap=cap)
    L.dic_ln_fwd(1, d["y2"].data_ptr(), gam.data_ptr(), gam.data_ptr(), d["h"].data_ptr(), d["mean"].data_ptr(), d["rstd"].data_ptr(), T, D, C.c_float(1e-12), st)

def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

T = 17408
full = mk(T); ha, hb = mk(T // 2), mk(T // 2)
s0 = torch.cuda.current_stream(); s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def one():
    for _ in range(12): chain(full, T, s0, 0)
def two(cap):
    def f():
        ev = torch.cuda.Event(); ev.record(s0); s1.wait_event(ev); s2.wait_event(ev)
        for _ in range(12):
            chain(ha, T // 2, s1, cap); chain(hb, T // 2, s2, cap)
        s0.wait_stream(s1); s0.wait_stream(s2)
    return f
print("one stream, full batch : %.3f ms per 12-layer forward chain" % timeit(one))
for cap in (0, 128, 160):
    print("two streams, halves, cap=%3d: %.3f ms" % (cap, timeit(two(cap))))

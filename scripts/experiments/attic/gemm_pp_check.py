#!/usr/bin/env python3
"""A/B of the two K loops of the bf16 256-column GEMM geometry (run on the GPU box): the ping-pong loop (csrc/gemm_pp.h, variant 1) against
the lock-step loop (variant 0) and against a float64 product -- every operand layout and epilogue, ragged M / N / K, one to many tiles per
workgroup (cu_cap forces a small persistent grid), K loops of 1, 2, 3, ... steps, split-K with the fused bias gradient, grouped weight
gradients.  `python scripts/gemm_pp_check.py` = correctness; `... time` = cold-operand timing of the step's shapes with both loops."""
import ctypes as C
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dic = importlib.import_module("diffusion-image-captioning_amd")
L = dic.lib()
GP = dic._lib.GemmParams
bf = torch.bfloat16
st = lambda: torch.cuda.current_stream().cuda_stream


def p(t):
    return 0 if t is None else t.data_ptr()


def run_gemm(variant, a_km, b_km, epi, **kw):
    L.dic_gemm_set_variant(variant)
    g = GP()
    for k, v in kw.items():
        setattr(g, k, v)
    rc = L.dic_gemm(1, a_km, b_km, epi, C.byref(g), st())
    assert rc == 0, L.dic_last_error().decode()
    torch.cuda.synchronize()


def relerr(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def case(M, N, K, a_km, b_km, epi=0, out_f32=0, resid=False, bias=True, p_drop=0.0, split=1, cu_cap=0, colsum=False, seed=0, tag=""):
    gen = torch.Generator().manual_seed(seed + M + 3 * N + 7 * K)
    A = (torch.randn(M, K, generator=gen) * 0.3)
    B = (torch.randn(N, K, generator=gen) * 0.3)
    Ad = (A.t().contiguous() if a_km else A).cuda().to(bf)
    Bd = (B.t().contiguous() if b_km else B).cuda().to(bf)
    Af = (Ad.float().t() if a_km else Ad.float()).double().cpu()
    Bf = (Bd.float().t() if b_km else Bd.float()).double().cpu()
    ref = Af @ Bf.t()
    biasd = torch.randn(N, generator=gen).cuda() if (bias and epi in (0, 1) and not out_f32 and split == 1) else None
    Rd = torch.randn(M, N, generator=gen).cuda().to(bf) if resid else None
    aux_in = torch.randn(M, N, generator=gen).cuda().to(bf) if epi == 2 else None
    outs = []
    for variant in (0, 1):
        Cd = torch.full((M + 3, N + 8), 7.0, dtype=torch.float32 if out_f32 else bf, device="cuda")
        aux = aux_in if epi == 2 else (torch.full((M + 3, N + 8), 7.0, dtype=bf, device="cuda") if epi == 1 else None)
        ws = torch.full((split * (M * N + M),), float("nan"), device="cuda") if split > 1 else None
        cs = torch.full((M,), float("nan"), device="cuda") if colsum else None
        kw = dict(A=p(Ad), B=p(Bd), C=p(Cd), M=M, N=N, K=K, lda=Ad.shape[1], ldb=Bd.shape[1], ldc=N + 8, tile=256, out_f32=out_f32,
                  bias=p(biasd), R=p(Rd), ldr=N, aux=p(aux), ldaux=(N if epi == 2 else N + 8), p_drop=p_drop, seed=1234, split_k=split,
                  split_ws=p(ws), cu_cap=cu_cap, colsum_out=p(cs))
        if split > 1:
            Cd = torch.full((M, N), 7.0, dtype=torch.float32, device="cuda")
            kw.update(C=p(Cd), ldc=N)
        run_gemm(variant, a_km, b_km, epi, **kw)
        outs.append((Cd.clone(), None if aux is None else aux.clone(), None if cs is None else cs.clone()))
    (c0, u0, s0), (c1, u1, s1) = outs
    name = f"{tag} M={M} N={N} K={K} km=({a_km},{b_km}) epi={epi} f32={out_f32} resid={int(resid)} drop={p_drop} split={split} cap={cu_cap}"
    # guards untouched
    if split == 1:
        assert bool((c1[:M, N:] == 7.0).all()) and bool((c1[M:] == 7.0).all()), "guard overwritten: " + name
    cc0, cc1 = (c0[:M, :N], c1[:M, :N]) if split == 1 else (c0, c1)
    d01 = relerr(cc1.float(), cc0.float())
    if epi == 0 and p_drop == 0.0:
        exp = ref + (biasd.double().cpu() if biasd is not None else 0) + (Rd.float().double().cpu() if resid else 0)
        e = relerr(cc1.float().cpu(), exp)
        assert e < (3e-6 if out_f32 else 8e-3), f"vs fp64 {e}: " + name
    if epi == 1:
        assert relerr(u1[:M, :N].float(), u0[:M, :N].float()) < 8e-3, "aux: " + name
        assert bool((u1[:M, N:] == 7.0).all()) and bool((u1[M:] == 7.0).all()), "aux guard: " + name
    if colsum:
        e = relerr(s1.cpu(), Af.sum(1))
        assert e < 3e-6, f"colsum {e}: " + name
    tol = 1e-6 if (out_f32 and biasd is None) else 1.6e-2
    assert d01 <= tol, f"variants differ {d01}: " + name
    print(f"ok  {name}   |pp-lockstep|={d01:.1e}", flush=True)


def group_case(T, cu_cap=0):
    gen = torch.Generator().manual_seed(T)
    shapes = [(768, 768), (2304, 768), (256, 264)]
    items_ref, res = [], []
    keep = []
    for variant in (0, 1):
        L.dic_gemm_set_variant(variant)
        gen = torch.Generator().manual_seed(T)
        arr = (dic._lib.WgradItem * len(shapes))()
        outs = []
        for i, (M, N) in enumerate(shapes):
            dY = (torch.randn(T, M, generator=gen) * 0.3).cuda().to(bf)
            X = (torch.randn(T, N, generator=gen) * 0.3).cuda().to(bf)
            dW = torch.full((M, N), float("nan"), device="cuda")
            db = torch.full((M,), float("nan"), device="cuda")
            keep += [dY, X]
            arr[i] = dic._lib.WgradItem(dY=p(dY), ldy=M, X=p(X), ldx=N, dW=p(dW), db=p(db), M=M, N=N)
            outs.append((dW, db, dY, X))
        nbytes = L.dic_wgrad_group_ws_bytes(arr, len(shapes), T, cu_cap)
        ws = torch.empty(nbytes // 4 + 4, device="cuda")
        rc = L.dic_wgrad_group(arr, len(shapes), T, p(ws), nbytes, cu_cap, st())
        assert rc == 0, L.dic_last_error().decode()
        torch.cuda.synchronize()
        res.append(outs)
    for (dW0, db0, dY, X), (dW1, db1, _, _) in zip(*res):
        ref = dY.float().double().cpu().t() @ X.float().double().cpu()
        assert relerr(dW1.cpu(), ref) < 3e-6 and relerr(db1.cpu(), dY.float().double().cpu().sum(0)) < 3e-6, f"group T={T}"
        assert relerr(dW1, dW0) < 1e-6 and relerr(db1, db0) < 1e-6
    print(f"ok  grouped weight gradients T={T} cap={cu_cap}", flush=True)


def correctness():
    # forward layouts, every epilogue, ragged everything; K loops of 1..5 steps
    for K in (64, 128, 192, 320, 768):
        case(600, 768, K, 0, 0, tag="fwd")
    case(600, 768, 320, 0, 0, resid=True, tag="fwd+resid")
    case(600, 768, 320, 0, 0, resid=True, p_drop=0.1, tag="fwd+resid+drop")
    case(600, 776, 192, 0, 0, out_f32=1, tag="fwd f32")
    case(509, 776, 128, 0, 0, epi=1, tag="gelu")
    case(509, 776, 128, 0, 0, epi=2, bias=False, tag="gelu'")
    # many tiles per workgroup (persistent grid of 3 / 5 workgroups), tile heights 4..8 fragments
    for M in (2048, 1800, 1500, 1200, 1000):
        case(M, 1024, 256, 0, 0, cu_cap=3, tag="persist")
    case(3000, 1536, 64, 0, 0, cu_cap=5, tag="persist nk=1")
    case(3000, 1536, 128, 0, 0, cu_cap=5, resid=True, tag="persist nk=2")
    case(3000, 1536, 192, 0, 1, cu_cap=5, tag="persist dX nk=3")
    # input gradients (k-major B)
    for K in (64, 320, 768):
        case(600, 768, K, 0, 1, tag="dX")
    case(600, 768, 320, 0, 1, resid=True, tag="dX+resid")
    case(600, 768, 256, 0, 1, epi=2, bias=False, tag="dX gelu'")
    case(2500, 768, 3072, 0, 1, out_f32=1, split=2, tag="dX split")
    # weight gradients (k-major A and B), ragged K, split-K, fused bias gradient
    for K in (64, 357, 64 * 23 + 17):
        case(512, 768, K, 1, 1, out_f32=1, bias=False, tag="dW")
    case(512, 768, 64 * 23 + 17, 1, 1, out_f32=1, bias=False, colsum=True, tag="dW+colsum")
    case(768, 520, 64 * 40 + 5, 1, 1, out_f32=1, bias=False, split=3, colsum=True, tag="dW split+colsum")
    case(768, 520, 64 * 9, 1, 1, out_f32=1, bias=False, split=5, colsum=True, cu_cap=4, tag="dW split persist")
    case(1024, 1024, 64 * 3, 1, 1, out_f32=1, bias=False, split=7, tag="dW split > nk: empty slices")
    for T, cap in ((64 * 11 + 5, 0), (64 * 40, 0), (64 * 3 + 7, 0), (64 * 70 + 1, 7), (64 * 30, 3)):
        group_case(T, cap)
    # rounding head
    M, V = 1024, 3000
    gen = torch.Generator().manual_seed(5)
    x = (torch.randn(M, 768, generator=gen) * 0.5).cuda().to(bf)
    W = (torch.randn(V, 768, generator=gen) * 0.05).cuda().to(bf)
    tgt = torch.randint(0, V, (M,), generator=gen).cuda()
    res = []
    for variant in (0, 1):
        npart = L.dic_ce_n_partials(V, 256)
        part = torch.zeros(M, npart, 4, device="cuda")
        tl = torch.zeros(M, device="cuda")
        run_gemm(variant, 0, 0, 3, A=p(x), B=p(W), C=0, M=M, N=V, K=768, lda=768, ldb=768, ldc=0, tile=256, tgt=p(tgt), partial=p(part), tgt_logit=p(tl), cu_cap=4)
        lse, am, nll = torch.empty(M, device="cuda"), torch.empty(M, dtype=torch.int64, device="cuda"), torch.empty(M, device="cuda")
        assert L.dic_ce_combine(p(part), p(tl), M, npart, p(lse), p(am), p(nll), st()) == 0
        vpad = (V + 127) // 128 * 128
        dl = torch.full((M, vpad), 7.0, dtype=bf, device="cuda")
        run_gemm(variant, 0, 0, 4, A=p(x), B=p(W), C=p(dl), M=M, N=V, K=768, lda=768, ldb=768, ldc=vpad, tile=256, tgt=p(tgt), lse=p(lse), ce_rows_a=M // 2,
                 ce_scale_a=0.5, ce_scale_b=0.25, cu_cap=4)
        res.append((lse.clone(), am.clone(), nll.clone(), dl.clone()))
    logits = x.float().double().cpu() @ W.float().double().cpu().t()
    assert relerr(res[1][0].cpu(), torch.logsumexp(logits, 1)) < 1e-5
    assert bool((res[1][1].cpu() == logits.argmax(1)).all())
    for a, b in zip(res[0], res[1]):
        assert bool((a == b).all()), "rounding head: the two loops differ"
    print("ok  rounding head (CE_PARTIAL, CE_DLOGITS): bit-identical between the loops", flush=True)
    L.dic_gemm_set_variant(1)
    print("ALL OK")


def timing():
    os.environ["COLD"] = "1"
    os.environ["TILE"] = "256"
    sys.argv = [sys.argv[0]]
    import runpy
    for variant in (1, 0, 1, 0):
        L.dic_gemm_set_variant(variant)
        print(f"==== variant {variant} ({'ping-pong' if variant else 'lock-step'})", flush=True)
        runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_bench.py"), run_name="__main__")
    L.dic_gemm_set_variant(1)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "time":
        timing()
    else:
        correctness()

"""GPU parity of the NARROW-tile asm GEMM (csrc/gemm_w4n.h, option gemm_w4n -- OFF by default).

These bodies were built in a round during which GPU use was closed to this repository: they are assembled by the real toolchain, reproduce numpy in instruction-level
emulation and are race-checked by symbolic execution (tests/test_host_cpu.py::test_narrow_tile_asm_gemm_is_generated_emulated_and_race_checked_on_the_cpu) but have not run
on an MI355X yet.  Until they have, this module is marked xfail(strict=False): its first hardware run is reported (XPASS / XFAIL in the log) without being able to turn the
suite of the SHIPPED path red -- the product never takes these kernels unless the option is set.  `scripts/experiments/w4n_ab.sh` runs the module with --runxfail (a
failure is a failure there).  Set HARDWARE_VERIFIED = True (and consider the option's default) once a log of that run is under profiles/.  The file name sorts last on
purpose: an experimental kernel runs after every test of the shipped path."""
import math

import pytest
import torch

from test_gpu_ops import BF16, L, _release_temporaries, dev, dic, gemm, p, relerr          # noqa: F401  (L, _release_temporaries: fixtures)

HARDWARE_VERIFIED = False
pytestmark = [pytest.mark.gpu] + ([] if HARDWARE_VERIFIED else
                                  [pytest.mark.xfail(strict=False, reason="narrow-tile asm GEMM: first hardware run pending (GPU use was closed while it was built)")])


@pytest.mark.parametrize("b_km", [0, 1])
@pytest.mark.parametrize("kind", ["plain", "bias", "resid", "bias_resid", "mulaux", "bias_resid_drop", "bias_gelu", "bias_gelud"])
@pytest.mark.parametrize("M,N,K,flat", [(256, 256, 576, 1), (768, 512, 768, 1), (768, 512, 768, 0), (4352, 768, 768, 1), (4352, 3072, 768, 1), (4352, 2304, 768, 0), (2304, 3072, 960, 1), (1024, 2304, 2304, 1)])
def test_gemm_narrow_tile_asm_kernel_is_bit_identical_to_the_wide_one(L, M, N, K, flat, kind, b_km):
    """dic_set_option("gemm_w4n", 1): launches the four-wave asm kernel accepts, with K a multiple of 192 in [576, gemm_w4n_kmax] and N a multiple of 128, run on
    the NARROW-tile bodies (csrc/gemm_w4n.h: 256 x 128 tiles, three LDS stages, the finished tile parked in a[128:255] while its epilogue is drained from the MFMA
    slots of the next tile's K loop).  Same MFMA order per accumulator, same epilogue arithmetic: the output must equal the wide bodies' BIT FOR BIT -- dropout
    mask, GELU and GELU' included (where K is not a multiple of 128 the wide bodies do not apply and the partner is the 8-wave kernel: 2 bf16 ulp) -- and float64 within
    bf16 rounding; guard rows behind C (and aux) must survive.  K = 576 / 768 / 960 / 2304: zero, one, two and 9 passes of the middle loop; M = 4352 x N = 768: 102 tiles on <= 256 workgroups (one tile each: prologue + post-loop epilogue only), 2304 x 3072: several
    tiles per workgroup (the overlapped path)."""
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K + b_km)
    A = torch.randn(M, K, generator=g) * 0.5
    W = torch.randn(N, K, generator=g) * 0.05
    bias = torch.randn(N, generator=g) if "bias" in kind else None
    side = torch.randn(M, N, generator=g) if ("resid" in kind or kind == "mulaux") else None
    Ad, Wd = dev(A, torch.bfloat16), dev(W.t() if b_km else W, torch.bfloat16)
    Sd = dev(side, torch.bfloat16) if side is not None else None
    exact = Ad.float().cpu().double() @ (Wd.float().cpu().double() if b_km else Wd.float().cpu().double().t())
    if bias is not None:
        exact = exact + bias.double()
    if side is not None:
        exact = exact * Sd.float().cpu().double() if kind == "mulaux" else exact + Sd.float().cpu().double()
    kw = dict(A=p(Ad), B=p(Wd), M=M, N=N, K=K, lda=K, ldb=(N if b_km else K), ldc=N, bias=p(dev(bias)) if bias is not None else 0, tile=256)
    if "resid" in kind:
        kw.update(R=p(Sd), ldr=N)
    if kind == "mulaux":
        kw.update(aux=p(Sd), ldaux=N)
    if "drop" in kind:
        if b_km:
            pytest.skip("dropout exists on the forward (k-contiguous B) launches only")
        kw.update(p_drop=0.1, seed=0xABCDEF0123 + M)
    epi = 7 if kind == "mulaux" else 0
    if "gelu" in kind:
        if b_km:
            pytest.skip("the GELU epilogues exist on the forward (k-contiguous B) launches only")
        epi = 6 if kind == "bias_gelud" else 1
        u = exact
        cdf = 0.5 * (1.0 + torch.erf(u / math.sqrt(2.0)))
        exact = u * cdf
    outs, auxs = [], []
    try:
        prev = L.dic_gemm_set_w4a(1)
        assert L.dic_set_option(b"gemm_w4a_mask", 0x3FF) == 0 and L.dic_set_option(b"gemm_w4n_mask", 0x3FF) == 0 and L.dic_set_option(b"gemm_w4n_kmax", 4096) == 0
        assert L.dic_set_option(b"gemm_w4n_flat", flat) == 0
        for narrow in (0, 1):
            Cfull = torch.full((M + 8, N), 7.0, dtype=torch.bfloat16, device="cuda")
            Cfull[:M].fill_(float("nan"))
            if kind == "bias_gelud":
                Afull = torch.full((M + 8, N), 5.0, dtype=torch.bfloat16, device="cuda")
                Afull[:M].fill_(float("nan"))
                kw.update(aux=p(Afull), ldaux=N)
                auxs.append(Afull)
            assert L.dic_set_option(b"gemm_w4n", narrow) == 0
            gemm(L, BF16, 0, b_km, epi, C=p(Cfull[:M]), **kw)
            torch.cuda.synchronize()
            outs.append(Cfull)
    finally:
        L.dic_gemm_set_w4a(prev)
        L.dic_set_option(b"gemm_w4n", 0)
        dic.options.push_to_library(L)
    assert bool((outs[1][M:] == 7.0).all()) and not bool(torch.isnan(outs[1][:M].float()).any())
    if K % 128 == 0:                  # the partner ran on the wide asm bodies
        assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)), "narrow and wide asm bodies differ"
    else:                             # K = 576 / 960: the wide bodies take pairs of K-steps, the partner was the 8-wave kernel (bias before the K loop: last-bit differences)
        d = (outs[1][:M].float() - outs[0][:M].float()).abs()
        if "drop" in kind:
            assert torch.equal(outs[0][:M].float() == Sd.float(), outs[1][:M].float() == Sd.float())       # the same dropout mask
        assert int((d > outs[0][:M].float().abs() * 2 ** -6 + 2e-3).sum()) == 0
    if kind == "bias_gelud":
        assert bool((auxs[1][M:] == 5.0).all())
        if K % 128 == 0:
            assert torch.equal(auxs[0].view(torch.int16), auxs[1].view(torch.int16))
        else:
            dd = (auxs[1][:M].float() - auxs[0][:M].float()).abs()
            assert int((dd > auxs[0][:M].float().abs() * 2 ** -6 + 2e-3).sum()) == 0
    if "drop" not in kind:
        assert relerr(outs[1][:M].float(), exact) < 6e-3


def test_narrow_tile_asm_kernel_refuses_nothing_silently(L):
    """Eligibility of the narrow bodies (gemm_w4n.h w4n_variant): K not a multiple of 192, K < 576 or above gemm_w4n_kmax keeps the wide bodies -- the output is the
    wide bodies' either way (this test pins that switching the option on can never change a result: no bias here, so the 8-wave kernel agrees bit for bit too), and
    gemm_w4n_kmax below 576 is refused."""
    M, N = 512, 256
    res = {}
    try:
        prev = L.dic_gemm_set_w4a(1)
        assert L.dic_set_option(b"gemm_w4n_mask", 0x7FF) == 0          # (the default mask leaves the plain form on the wide bodies)
        for K in (256, 384, 640, 768, 1152):
            g = torch.Generator().manual_seed(K)
            Ad, Wd = dev(torch.randn(M, K, generator=g) * 0.5, torch.bfloat16), dev(torch.randn(N, K, generator=g) * 0.05, torch.bfloat16)
            for narrow in (0, 1):
                Cd = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
                assert L.dic_set_option(b"gemm_w4n", narrow) == 0
                gemm(L, BF16, 0, 0, 0, A=p(Ad), B=p(Wd), C=p(Cd), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, tile=256)
                res[(K, narrow)] = Cd
            assert torch.equal(res[(K, 0)], res[(K, 1)]), K
        assert L.dic_set_option(b"gemm_w4n_kmax", 512) != 0
    finally:
        L.dic_gemm_set_w4a(prev)
        L.dic_set_option(b"gemm_w4n", 0)
        dic.options.push_to_library(L)


@pytest.mark.parametrize("M,V,flat", [(512, 3000, 1), (4352, 30522, 1), (1024, 30522, 0)])
def test_rounding_head_forward_on_the_narrow_tile_kernel_matches_the_eight_wave_kernel(L, M, V, flat):
    """DIC_EPI_CE_EXP (training forward of the rounding loss: E = bf16(exp(logit - c_row)), zeros in columns [V, ldE), unrounded sums per 64-column slab, the target's
    logit) on the narrow bodies (gemm_w4n.h launch_w4n_ce) against the 8-wave kernel on the same operands.  The 8-wave kernel starts its accumulators from the bias, the
    narrow body adds it behind the K loop, and the lanes add a row's 64 values up in another order: E within 2 bf16 ulp, sums to 1e-5, target logits to 1e-4 absolute.
    Rows with a target outside [0, V) must keep their tgt_logit; the slab slots beyond ldE (cleared by the launcher) must be zero; guard rows must survive."""
    g = torch.Generator().manual_seed(M + V)
    K, ldE = 768, (V + 127) // 128 * 128
    npart = L.dic_ce_n_partials(V, 256)
    X = dev(torch.randn(M, K, generator=g) * 0.5, torch.bfloat16)
    W = dev(torch.randn(V, K, generator=g) * 0.05, torch.bfloat16)
    bias = dev(torch.randn(V + 256, generator=g) * 0.1)
    logits = X.float() @ W.float().t() + bias[:V]
    cref = dev((logits.max(dim=1).values - 2.0).contiguous())
    tgt = torch.randint(0, V, (M,), generator=g)
    tgt[1], tgt[2], tgt[M - 1] = -1, V + 5, V - 1
    tgt = dev(tgt)
    res = []
    try:
        prev = L.dic_gemm_set_w4a(1)
        assert L.dic_set_option(b"gemm_w4n_mask", 0x7FF) == 0 and L.dic_set_option(b"gemm_w4n_flat", flat) == 0
        for narrow in (0, 1):
            E = torch.full((M + 8, ldE), 7.0, dtype=torch.bfloat16, device="cuda")
            part = torch.full((M + 8, npart), 3.0, dtype=torch.float32, device="cuda")
            tl = torch.full((M + 8,), 9.0, dtype=torch.float32, device="cuda")
            assert L.dic_set_option(b"gemm_w4n", narrow) == 0
            gemm(L, BF16, 0, 0, dic._lib.EPI_CE_EXP, A=p(X), B=p(W), C=p(E), M=M, N=V, K=K, lda=K, ldb=K, ldc=ldE, bias=p(bias), lse=p(cref), tgt=p(tgt), partial=p(part), tgt_logit=p(tl), tile=256)
            torch.cuda.synchronize()
            res.append((E, part, tl))
    finally:
        L.dic_gemm_set_w4a(prev)
        L.dic_set_option(b"gemm_w4n", 0)
        dic.options.push_to_library(L)
    (E0, P0, T0), (E1, P1, T1) = res
    assert bool((E1[M:] == 7.0).all()) and bool((P1[M:] == 3.0).all()) and bool((T1[M:] == 9.0).all())
    assert bool((E1[:M, V:] == 0).all()) and bool((P1[:M, 2 * (ldE // 128):] == 0).all())
    d = (E1[:M, :V].float() - E0[:M, :V].float()).abs()
    assert int((d > E0[:M, :V].float().abs() * 2 ** -6 + 1e-6).sum()) == 0
    assert relerr(P1[:M], P0[:M]) < 1e-5
    valid = (tgt >= 0) & (tgt < V)
    assert float((T1[:M][valid] - T0[:M][valid]).abs().max()) < 1e-4 and bool((T1[:M][~valid] == 9.0).all())
    want = torch.exp(logits.double().cpu() - cref.double().cpu()[:, None])
    assert relerr(E1[:M, :V].float(), want) < 6e-3


def test_training_steps_with_the_narrow_tile_kernel_on_every_eligible_launch_match_the_shipped_configuration(L):
    """End to end: two AdamW steps of the default engine (12 288 token rows = 48 x 256, 2 layers, the full 30 522-word rounding head, dropout ON) with options.gemm_w4n on
    -- q|k|v, out-proj, FFN lin1 (+ GELU'), the K = 768 input gradients and the rounding-head forward then run on the narrow bodies -- against the same steps in the
    shipped configuration.  Where the shipped step runs the wide asm bodies the narrow ones are bit-identical; the dropout + residual, GELU + GELU' and CE_EXP launches move
    from the 8-wave kernel (bias in front of the K loop, another order of a row's slab sum): the four losses agree to 2e-5 and the parameters after two steps to 1e-5 of their
    norm -- same dropout masks (one hash), same targets.  This is the test that exercises launch_w4n_ce inside the engine (np, ldE = vpad, the cleared slab slots)."""
    import importlib
    import numpy as np
    opts = importlib.import_module("diffusion-image-captioning_amd.options")
    synth = dic.synth
    B, S, Lc, V, nl = 128, 1, 16, 30522, 2
    dic.cfg.update(BATCH_SIZE=B, SAMPLE_SIZE=S, MAX_LENGTH=Lc, STEP_TOT=100, COSIN_SCHEDULE=False, ROUNDING_WEIGHT=0.5, VOCAB_SIZE=V,
                   LOSS_FUNC="series_sum_sample_mean", CLIP_ADDING_METHOD="concat", CLASSIFIER_FREE_WEIGHT=0.0, X_0_PREDICTION=True)
    dic.set_alpha_cumprod(None)
    E = synth.vocab_embedding(V, 768, 0)
    state = synth.denoiser_state(nl, 0)
    x = {k: torch.from_numpy(v).cuda() for k, v in synth.batch(B, Lc, V, 1).items()}
    t = torch.from_numpy(synth.timesteps(S, 100, 0))
    nz = [torch.from_numpy(synth.noise((B, Lc, 768), 3, f"eps{i}")) for i in range(2)]
    out = []
    try:
        for narrow in (False, True):
            opts.set_option("gemm_w4n", narrow)
            opts.set_option("gemm_w4n_mask", 0x7FF)
            model = dic.DistilBertModel(E, E, config=dict(n_layers=nl, dropout=0.1, attention_dropout=0.1), dtype="bf16", seed=0)
            model.load_state(state)
            trainer = dic.AdamW(model.parameters(), lr=1e-4)
            losses = []
            for _ in range(2):
                losses.append([float(v) for v in dic.train_func(model, trainer, x, t=t, noises=nz)])
            torch.cuda.synchronize()
            # (k_lin.bias has an analytically zero gradient -- softmax shift invariance -- and Adam turns its round-off into +-lr noise: left out, as in smoke())
            out.append((np.array(losses), [p_.detach().float().clone() for n_, p_ in model.named_parameters() if not n_.endswith("k_lin.bias")]))
            del model, trainer
            torch.cuda.empty_cache()
    finally:
        opts.set_option("gemm_w4n", False)
        opts.set_option("gemm_w4n_mask", opts.Options().gemm_w4n_mask)
        dic.options.push_to_library(L)
    (l0, p0), (l1, p1) = out
    assert np.isfinite(l1).all() and (np.abs(l1 - l0) <= 2e-5 * np.abs(l0)).all(), (l0, l1)
    worst = max(float((a - b).norm() / (a.norm() + 1e-30)) for a, b in zip(p0, p1))
    assert worst < 1e-5, worst

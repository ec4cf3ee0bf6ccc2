"""Per-kernel parity: every C-ABI entry point against the CPU oracle (oracle/ref_model.py, oracle/rounding.c) or a
float64 restatement, on seeded inputs, through ctypes -- the same calls the product makes.  GPU only."""
import ctypes as C
import importlib
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

dic = importlib.import_module("diffusion-image-captioning_amd")
from oracle import ref_model as R                      # noqa: E402
from oracle.rounding import rounding_ref               # noqa: E402

F32, BF16 = 0, 1
DT = {F32: torch.float32, BF16: torch.bfloat16}


@pytest.fixture(scope="module")
def L():
    dic._lib.require_gpu()
    return dic.lib()


_KEEP = []          # device temporaries must outlive the (asynchronous) kernel that reads them


@pytest.fixture(autouse=True)
def _release_temporaries():
    yield
    torch.cuda.synchronize()
    _KEEP.clear()


def dev(t, dt=None):
    t = torch.as_tensor(t)
    d = t.to("cuda", dt if dt is not None else t.dtype).contiguous()
    _KEEP.append(d)
    return d


def p(t):
    return 0 if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def ok(rc, L):
    assert rc == 0, L.dic_last_error().decode()


def relerr(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-30))


def gemm(L, dtype, a_km, b_km, epi, **kw):
    g = dic._lib.GemmParams()
    for k, v in kw.items():
        setattr(g, k, v)
    ok(L.dic_gemm(dtype, a_km, b_km, epi, C.byref(g), stream()), L)
    torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------------ layout probe
def test_tr16_layout_assumption(L):
    src = torch.arange(256, dtype=torch.int16, device="cuda")
    out = torch.zeros(256, dtype=torch.int16, device="cuda")
    ok(L.dic_probe_tr16(p(src), p(out), stream()), L)
    got = out.cpu().numpy().reshape(64, 4)
    exp = np.array([[(l & 15) + 16 * j + 64 * (l >> 4) for j in range(4)] for l in range(64)])
    assert (got == exp).all(), f"ds_read_b64_tr_b16 mapping differs:\n{got[:20]}"


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("layout", [(0, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("shape", [(200, 256, 256), (128, 128, 64), (392, 384, 192)])
def test_gemm_layouts(L, dtype, layout, shape):
    a_km, b_km = layout
    M, N, K = shape
    if a_km:
        M = (M + 7) // 8 * 8
        K = K + 37                       # ragged contraction (tokens) is legal when both operands are k-major
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + dtype + 10 * a_km + 20 * b_km)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g)
    Aq, Bq = A.to(DT[dtype]).float(), B.to(DT[dtype]).float()
    ref = Aq.double() @ Bq.double().t()
    Ad = dev(A.t() if a_km else A, DT[dtype])
    Bd = dev(B.t() if b_km else B, DT[dtype])
    Cd = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
    gemm(L, dtype, a_km, b_km, 0, A=p(Ad), B=p(Bd), C=p(Cd), M=M, N=N, K=K, lda=Ad.shape[1], ldb=Bd.shape[1], ldc=N, out_f32=1)
    e = relerr(Cd, ref)
    assert e < (2e-6 if dtype == F32 else 2e-6), f"relerr {e}"   # inputs pre-rounded: both accumulate in fp32


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("split", [2, 5])
def test_gemm_split_k_weight_gradient(L, dtype, split):
    """dW = dY^T X with the contraction (tokens) cut into slices that are folded in a fixed order."""
    M, N, K = 256, 384, 64 * 23 + 17
    g = torch.Generator().manual_seed(split)
    A, B = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
    Ad, Bd = dev(A, DT[dtype]), dev(B, DT[dtype])
    ref = Ad.float().cpu().double().t() @ Bd.float().cpu().double()
    Cd = torch.full((M, N), 1.0, dtype=torch.float32, device="cuda")
    ws = torch.full((split * M * N,), float("nan"), device="cuda")
    gemm(L, dtype, 1, 1, 0, A=p(Ad), B=p(Bd), C=p(Cd), M=M, N=N, K=K, lda=M, ldb=N, ldc=N, out_f32=1, split_k=split, split_ws=p(ws))
    assert relerr(Cd, ref) < 2e-6
    gemm(L, dtype, 1, 1, 0, A=p(Ad), B=p(Bd), C=p(Cd), M=M, N=N, K=K, lda=M, ldb=N, ldc=N, out_f32=1, split_k=split, split_ws=p(ws), accumulate=1)
    assert relerr(Cd, 2 * ref) < 2e-6


@pytest.mark.parametrize("layout", [(0, 0), (0, 1), (1, 1)])
def test_gemm_tile256_all_layouts_and_epilogues(L, layout):
    """The 256x256 / 8-wave geometry of the bf16 kernel: ragged M, N = 3 tiles, bias + residual, GELU pair, split-K + fused bias grad."""
    a_km, b_km = layout
    M, N, K = (512 if a_km else 600), 768, 320 + (37 if (a_km and b_km) else 0)
    g = torch.Generator().manual_seed(7 + a_km + 2 * b_km)
    A, B = torch.randn(M, K, generator=g) * 0.3, torch.randn(N, K, generator=g) * 0.3
    bias, Rr = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    Ad, Bd, Rd = dev(A.t() if a_km else A, torch.bfloat16), dev(B.t() if b_km else B, torch.bfloat16), dev(Rr, torch.bfloat16)
    acc = (Ad.float().cpu().double().t() if a_km else Ad.float().cpu().double()) @ (Bd.float().cpu().double() if b_km else Bd.float().cpu().double().t())
    lda, ldb = Ad.shape[1], Bd.shape[1]
    Cd = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    gemm(L, BF16, a_km, b_km, 0, A=p(Ad), B=p(Bd), C=p(Cd), M=M, N=N, K=K, lda=lda, ldb=ldb, ldc=N, bias=p(dev(bias)), R=p(Rd), ldr=N, tile=256)
    assert relerr(Cd.float(), acc + bias.double() + Rd.float().cpu().double()) < 6e-3
    Cf = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
    gemm(L, BF16, a_km, b_km, 0, A=p(Ad), B=p(Bd), C=p(Cf), M=M, N=N, K=K, lda=lda, ldb=ldb, ldc=N, out_f32=1, tile=256)
    assert relerr(Cf, acc) < 2e-6
    if not a_km:
        U = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        G_ = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        gemm(L, BF16, 0, b_km, 1, A=p(Ad), B=p(Bd), C=p(G_), M=M, N=N, K=K, lda=lda, ldb=ldb, ldc=N, bias=p(dev(bias)), aux=p(U), ldaux=N, tile=256)
        assert relerr(U.float(), acc + bias.double()) < 1e-2 and relerr(G_.float(), R.gelu(U.float().cpu().double())) < 1e-2
        Dd = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        gemm(L, BF16, 0, b_km, 2, A=p(Ad), B=p(Bd), C=p(Dd), M=M, N=N, K=K, lda=lda, ldb=ldb, ldc=N, aux=p(U), ldaux=N, tile=256)
        uu = U.float().cpu().double().requires_grad_(True)
        R.gelu(uu).sum().backward()
        assert relerr(Dd.float(), acc * uu.grad) < 1e-2
    else:
        for split in (1, 3):
            cs = torch.full((M,), float("nan"), device="cuda")
            ws = torch.zeros(split * (M * N + M), device="cuda")
            Cs = torch.zeros(M, N, dtype=torch.float32, device="cuda")
            gemm(L, BF16, 1, 1, 0, A=p(Ad), B=p(Bd), C=p(Cs), M=M, N=N, K=K, lda=lda, ldb=ldb, ldc=N, out_f32=1, split_k=split, split_ws=p(ws),
                 colsum_out=p(cs), tile=256)
            assert relerr(Cs, acc) < 2e-6 and relerr(cs, Ad.float().cpu().double().sum(0)) < 2e-6


@pytest.mark.parametrize("tile,M,N,K,epi", [(256, 600, 768, 768, 0), (256, 17408 // 8, 3072, 768, 1), (128, 300, 776, 64, 0), (256, 520, 768, 3072, 0)])
def test_gemm_split_weight_second_pass(L, tile, M, N, K, epi):
    """DicGemmParams.B2: C = A (B + B2)^T, the forward Linear against hi + lo bf16 halves of an fp32 weight (hf nn.Linear with its fp32 weight,
    hf:183-185, 201, 221-223, 510), as two passes of the K loop.  Checked against float64 on the SAME operands (A bf16, W_hi + W_lo): the only
    error left is the fp32 accumulation; and against the fp32 weight itself: 2^-16, where the single-pass bf16 product is 2^-8 away."""
    g = torch.Generator().manual_seed(31 + K)
    A = torch.randn(M, K, generator=g) * 0.5
    W = torch.randn(N, K, generator=g) * 0.03
    bias, Rr = torch.randn(N, generator=g) * 0.1, torch.randn(M, N, generator=g)
    Ad, Wd = dev(A, torch.bfloat16), dev(W)
    hi, lo = torch.zeros(N, K, dtype=torch.bfloat16, device="cuda"), torch.full((N, K), 7.0, dtype=torch.bfloat16, device="cuda")
    ok(L.dic_cast_bf16_hl(p(Wd), p(hi), p(lo), N * K, stream()), L)
    torch.cuda.synchronize()
    assert torch.equal(hi.cpu(), W.to(torch.bfloat16)) and torch.equal(lo.cpu(), (W - W.to(torch.bfloat16).float()).to(torch.bfloat16))
    A64 = Ad.float().cpu().double()
    exact = A64 @ (hi.float().cpu().double() + lo.float().cpu().double()).t() + bias.double()
    full = A64 @ W.double().t() + bias.double()
    if epi == 0:
        Rd = dev(Rr, torch.bfloat16)
        Cf = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
        gemm(L, BF16, 0, 0, 0, A=p(Ad), B=p(hi), B2=p(lo), C=p(Cf), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=p(dev(bias)), out_f32=1, tile=tile)
        assert relerr(Cf, exact) < 2e-6
        assert relerr(Cf, full) < 3e-5, "hi + lo must carry ~16 mantissa bits of the fp32 weight"
        C1 = torch.zeros(M, N, dtype=torch.float32, device="cuda")
        gemm(L, BF16, 0, 0, 0, A=p(Ad), B=p(hi), C=p(C1), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=p(dev(bias)), out_f32=1, tile=tile)
        assert relerr(C1, full) > 10 * relerr(Cf, full)            # what the second pass buys
        Cb = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")   # the engine's form: bf16 output with residual
        gemm(L, BF16, 0, 0, 0, A=p(Ad), B=p(hi), B2=p(lo), C=p(Cb), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=p(dev(bias)), R=p(Rd), ldr=N, tile=tile)
        assert relerr(Cb.float(), exact + Rd.float().cpu().double()) < 6e-3
    else:
        U = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        G_ = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        gemm(L, BF16, 0, 0, 1, A=p(Ad), B=p(hi), B2=p(lo), C=p(G_), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=p(dev(bias)), aux=p(U), ldaux=N, tile=tile)
        assert torch.equal(U.cpu(), exact.float().to(torch.bfloat16)) or relerr(U.float(), exact) < 4e-3
        assert relerr(G_.float(), R.gelu(exact)) < 8e-3
    if epi == 0 and N % 256 == 0:
        # b2_col0: only the column tiles from b2_col0 on take the second pass (the value third of a fused q|k|v weight); the others are B alone
        c0 = 512
        Cp = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
        gemm(L, BF16, 0, 0, 0, A=p(Ad), B=p(hi), B2=p(lo), b2_col0=c0, C=p(Cp), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=p(dev(bias)), out_f32=1, tile=tile)
        assert torch.equal(Cp[:, :c0], C1[:, :c0]) and torch.equal(Cp[:, c0:], Cf[:, c0:])
        gq = dic._lib.GemmParams()
        for k, v in dict(A=p(Ad), B=p(hi), B2=p(lo), b2_col0=100, C=p(Cp), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, out_f32=1, tile=tile).items():
            setattr(gq, k, v)
        assert L.dic_gemm(BF16, 0, 0, 0, C.byref(gq), stream()) != 0             # not a multiple of the tile width
    # refused where it is not built: k-major operands, split-K
    gp = dic._lib.GemmParams()
    for k, v in dict(A=p(Ad), B=p(hi), B2=p(lo), C=p(hi), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, tile=tile).items():
        setattr(gp, k, v)
    assert L.dic_gemm(BF16, 0, 1, 0, C.byref(gp), stream()) != 0


@pytest.mark.parametrize("rows", [256, 224])
@pytest.mark.parametrize("b_km", [0, 1])
@pytest.mark.parametrize("kind", ["plain", "bias", "resid", "bias_resid", "mulaux", "bias_resid_drop", "bias_gelu", "bias_gelud"])
@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (768, 512, 384), (2304, 768, 3072), (4352, 2304, 768)])
def test_gemm_four_wave_asm_kernel_matches_float64_and_the_default_kernel(L, M, N, K, kind, b_km, rows):
    """dic_gemm_set_w4a(1): eligible launches (bf16, k-contiguous A, M and N multiples of 256, K of 128; AFFINE with optional bias / residual, MUL_AUX)
    run on the hand-scheduled four-wave kernel (csrc/gemm_w4a.h: one generated asm statement per variant).  Every variant against float64 on the
    same operands and against the 8-wave kernel (which adds the bias before the K loop instead of after it: differences of at most 2 bf16 ulp);
    launches outside its scope (dropout, fp32 output, ragged sizes) must fall through to the 8-wave kernel unchanged.
    rows: both generated tile heights (dic_set_option gemm_w4a_rows; 0 = chosen per launch).  With 224-row tiles none of the M here is a multiple of the
    tile height: the last row tile is ragged (32 ... 160 valid rows), its rows >= M must read as zeros and must not be stored (guard rows behind C)."""
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
    auxs = []
    if "gelu" in kind:        # FFN lin1 forward: C = GELU(u); "gelud" (training): also aux = GELU'(u), u = acc + bias
        if b_km:
            pytest.skip("the GELU epilogues exist on the forward (k-contiguous B) launches only")
        epi = 6 if kind == "bias_gelud" else 1
        u = exact
        cdf = 0.5 * (1.0 + torch.erf(u / math.sqrt(2.0)))
        exact_d = cdf + u * torch.exp(-0.5 * u * u) / math.sqrt(2.0 * math.pi)
        exact = u * cdf
    outs = []
    for mode in (0, 1):
        Cfull = torch.full((M + 8, N), 7.0, dtype=torch.bfloat16, device="cuda")
        Cd = Cfull[:M]
        Cd.fill_(float("nan"))
        if kind == "bias_gelud":
            Afull = torch.full((M + 8, N), 5.0, dtype=torch.bfloat16, device="cuda")
            Afull[:M].fill_(float("nan"))
            kw.update(aux=p(Afull), ldaux=N)
            auxs.append(Afull)
        prev = L.dic_gemm_set_w4a(mode)
        assert L.dic_set_option(b"gemm_w4a_rows", rows) == 0 and L.dic_set_option(b"gemm_w4a_mask", 0x3FF) == 0
        try:
            gemm(L, BF16, 0, b_km, epi, C=p(Cd), **kw)
        finally:
            L.dic_gemm_set_w4a(prev)
            dic.options.push_to_library(L)
        assert bool((Cfull[M:] == 7.0).all())
        outs.append(Cd)
    if kind == "bias_gelud":
        assert bool((auxs[0][M:] == 5.0).all()) and bool((auxs[1][M:] == 5.0).all())
        assert relerr(auxs[1][:M].float(), exact_d) < 6e-3
        dd = (auxs[1][:M].float() - auxs[0][:M].float()).abs()
        assert int((dd > auxs[0][:M].float().abs() * 2 ** -6 + 2e-3).sum()) == 0
    if "drop" in kind:
        # the mask must be the 8-wave kernel's (ln_bwd regenerates it from the same hash): identical zero pattern of (out - R), kept elements scaled by 1 / 0.9
        kept0, kept1 = (outs[0].float() - Sd.float()) != 0, (outs[1].float() - Sd.float()) != 0
        assert torch.equal(kept0, kept1) and abs(float((~kept1).float().mean()) - 0.1) < 0.01
        exact = torch.where(kept1.cpu(), (exact - Sd.float().cpu().double()) * (65536.0 / (65536.0 - 6554.0)) + Sd.float().cpu().double(), Sd.float().cpu().double())
    assert relerr(outs[1].float(), exact) < 6e-3
    d = (outs[1].float() - outs[0].float()).abs()
    assert int((d > outs[0].float().abs() * 2 ** -6 + 2e-3).sum()) == 0
    if kind == "plain" and not b_km:          # out of scope: falls through, bit-identical to the default path
        for extra in (dict(p_drop=0.25, seed=5), dict(out_f32=1)):
            res = []
            for mode in (0, 1):
                Cx = torch.zeros(M, N, dtype=torch.float32 if "out_f32" in extra else torch.bfloat16, device="cuda")
                prev = L.dic_gemm_set_w4a(mode)
                try:
                    gemm(L, BF16, 0, 0, 0, C=p(Cx), **kw, **extra)
                finally:
                    L.dic_gemm_set_w4a(prev)
                res.append(Cx)
            assert torch.equal(res[0], res[1])


@pytest.mark.parametrize("tile", [128, 256])
def test_gemm_bf16_epilogue_general_path(L, tile):
    """The rarely used epilogue combinations that need loads inside the row loop: accumulate into an fp32 C, and a residual
    with N % 8 == 4 (the last 8-column item of a row is half valid)."""
    M, N, K = 300, 772, 192
    g = torch.Generator().manual_seed(tile)
    A, B = torch.randn(M, K, generator=g) * 0.3, torch.randn(N, K, generator=g) * 0.3
    bias, Rr = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    Ad, Bd, Rd = dev(A, torch.bfloat16), dev(B, torch.bfloat16), dev(Rr, torch.bfloat16)
    acc = Ad.float().cpu().double() @ Bd.float().cpu().double().t()
    Cd = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    gemm(L, BF16, 0, 0, 0, A=p(Ad), B=p(Bd), C=p(Cd), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=p(dev(bias)), R=p(Rd), ldr=N, tile=tile)
    assert relerr(Cd.float(), acc + bias.double() + Rd.float().cpu().double()) < 6e-3
    Cf = torch.full((M, N), 2.0, dtype=torch.float32, device="cuda")
    gemm(L, BF16, 0, 0, 0, A=p(Ad), B=p(Bd), C=p(Cf), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, out_f32=1, accumulate=1, tile=tile)
    assert relerr(Cf, acc + 2.0) < 2e-6
    # and the fast path on the same ragged N without a residual: nothing may be written beyond column N-1 of a row
    Cp = torch.full((M, N + 4), 7.0, dtype=torch.bfloat16, device="cuda")
    gemm(L, BF16, 0, 0, 0, A=p(Ad), B=p(Bd), C=p(Cp), M=M, N=N, K=K, lda=K, ldb=K, ldc=N + 4, tile=tile)
    assert relerr(Cp[:, :N].float(), acc) < 6e-3 and bool((Cp[:, N:] == 7.0).all())


@pytest.mark.parametrize("tile", [128, 256])
@pytest.mark.parametrize("M", [301, 509])
def test_gemm_line_store_edges(L, tile, M):
    """The epilogue writes whole cache lines by swapping a chunk between lanes t and t^8 (rows r and r+8 of a fragment): the last
    valid row falls inside an 8-row half (M % 8 = 5), the fp32 output has N % 8 == 4 (the line's last 16-byte chunk is the
    row's last), every buffer has guard columns / rows that must keep their fill value."""
    N, K = 776, 128
    g = torch.Generator().manual_seed(M + tile)
    A, B = torch.randn(M, K, generator=g) * 0.3, torch.randn(N, K, generator=g) * 0.3
    bias, Rr = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    Ad, Bd, Rd = dev(A, torch.bfloat16), dev(B, torch.bfloat16), dev(Rr, torch.bfloat16)
    acc = Ad.float().cpu().double() @ Bd.float().cpu().double().t()
    Cd = torch.full((M + 9, N + 8), 7.0, dtype=torch.bfloat16, device="cuda")
    gemm(L, BF16, 0, 0, 0, A=p(Ad), B=p(Bd), C=p(Cd), M=M, N=N, K=K, lda=K, ldb=K, ldc=N + 8, bias=p(dev(bias)), R=p(Rd), ldr=N, tile=tile)
    assert relerr(Cd[:M, :N].float(), acc + bias.double() + Rd.float().cpu().double()) < 6e-3
    assert bool((Cd[:M, N:] == 7.0).all()) and bool((Cd[M:] == 7.0).all())
    N2 = 772
    Cf = torch.full((M + 9, N2 + 4), 7.0, dtype=torch.float32, device="cuda")
    gemm(L, BF16, 0, 0, 0, A=p(Ad), B=p(Bd), C=p(Cf), M=M, N=N2, K=K, lda=K, ldb=K, ldc=N2 + 4, out_f32=1, tile=tile)
    assert relerr(Cf[:M, :N2], acc[:, :N2]) < 2e-6
    assert bool((Cf[:M, N2:] == 7.0).all()) and bool((Cf[M:] == 7.0).all())
    U = torch.full((M + 9, N + 8), 7.0, dtype=torch.bfloat16, device="cuda")
    G_ = torch.full((M + 9, N + 8), 7.0, dtype=torch.bfloat16, device="cuda")
    gemm(L, BF16, 0, 0, 1, A=p(Ad), B=p(Bd), C=p(G_), M=M, N=N, K=K, lda=K, ldb=K, ldc=N + 8, bias=p(dev(bias)), aux=p(U), ldaux=N + 8, tile=tile)
    assert relerr(U[:M, :N].float(), acc + bias.double()) < 1e-2 and relerr(G_[:M, :N].float(), R.gelu(U[:M, :N].float().cpu().double())) < 1e-2
    assert bool((U[:M, N:] == 7.0).all()) and bool((U[M:] == 7.0).all()) and bool((G_[:M, N:] == 7.0).all()) and bool((G_[M:] == 7.0).all())
    Dd = torch.full((M + 9, N + 8), 7.0, dtype=torch.bfloat16, device="cuda")
    gemm(L, BF16, 0, 0, 2, A=p(Ad), B=p(Bd), C=p(Dd), M=M, N=N, K=K, lda=K, ldb=K, ldc=N + 8, aux=p(U), ldaux=N + 8, tile=tile)
    uu = U[:M, :N].float().cpu().double().requires_grad_(True)
    R.gelu(uu).sum().backward()
    assert relerr(Dd[:M, :N].float(), acc * uu.grad) < 1e-2
    assert bool((Dd[:M, N:] == 7.0).all()) and bool((Dd[M:] == 7.0).all())


@pytest.mark.parametrize("tile,M", [(128, 301), (256, 509), (256, 1792)])
@pytest.mark.parametrize("p_drop", [0.0, 0.1])
def test_gemm_fp32_residual_stream_epilogue(L, tile, M, p_drop):
    """out_f32 = DIC_OUT_F32 | DIC_RES_IS_F32 (include/dic_hip.h, DIC_RES_F32): C = dropout(A B^T + bias) + R with C and R in fp32 -- the residual
    GEMMs of a block in the fp32-residual-stream mode.  Against float64 on the bf16-rounded operands to fp32 accuracy (the bf16-residual form is
    only good to 4e-3), ragged last rows, guard rows / columns untouched; the dropout mask is the one the bf16 epilogue draws (same seed, same
    (row, column) keys): the kept elements agree with the bf16-output launch."""
    N, K = 768, 192
    g = torch.Generator().manual_seed(M + tile)
    A, B = torch.randn(M, K, generator=g) * 0.3, torch.randn(N, K, generator=g) * 0.3
    bias, Rr = torch.randn(N, generator=g), torch.randn(M, N, generator=g) * 3
    Ad, Bd, Rd = dev(A, torch.bfloat16), dev(B, torch.bfloat16), dev(Rr)
    acc = Ad.float().cpu().double() @ Bd.float().cpu().double().t() + bias.double()
    Cf = torch.full((M + 9, N + 8), 7.0, dtype=torch.float32, device="cuda")
    gemm(L, BF16, 0, 0, 0, A=p(Ad), B=p(Bd), C=p(Cf), M=M, N=N, K=K, lda=K, ldb=K, ldc=N + 8, bias=p(dev(bias)), R=p(Rd), ldr=N, out_f32=3, tile=tile,
         p_drop=p_drop, seed=77)
    assert bool((Cf[:M, N:] == 7.0).all()) and bool((Cf[M:] == 7.0).all())
    got = Cf[:M, :N].cpu().double() - Rr.double()
    if p_drop == 0.0:
        assert relerr(Cf[:M, :N], acc + Rr.double()) < 2e-6
    else:
        Cb = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        gemm(L, BF16, 0, 0, 0, A=p(Ad), B=p(Bd), C=p(Cb), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=p(dev(bias)), tile=tile, p_drop=p_drop, seed=77)
        keep = Cb.float().cpu() != 0
        assert 0.85 < keep.float().mean() < 0.95
        assert float((got[~keep]).abs().max()) < 2e-5                                   # dropped: only the residual is left
        assert float(((got - acc / (1 - p_drop))[keep]).abs().max()) < 2e-5 * float(acc.abs().max())


@pytest.mark.parametrize("T,K,N,stride", [(17408, 768, 2304, 16), (17408, 3072, 768, 16), (300, 768, 772, 1), (1000, 3072, 3072, 7)])
def test_lo_mean_bias_is_bias_plus_lo_times_the_mean_of_the_sampled_rows(L, T, K, N, stride):
    """dic_lo_mean_bias: bias_eff = bias + lo . mean(A[0::stride]) -- the row-common part of A W_lo^T, handed to the GEMM as its bias instead of
    a second pass of the K loop (include/dic_hip.h).  Against float64 on the same bf16 operands; a second call gives the same bits."""
    g = torch.Generator().manual_seed(T + K + N)
    A = torch.randn(T, K, generator=g) * 0.7 + torch.randn(1, K, generator=g) * 0.4            # rows with a common part, like LayerNorm outputs
    W = torch.randn(N, K, generator=g) * 0.03
    bias = torch.randn(N, generator=g) * 0.1
    Ad, Wd = dev(A, torch.bfloat16), dev(W)
    hi, lo = torch.zeros(N, K, dtype=torch.bfloat16, device="cuda"), torch.zeros(N, K, dtype=torch.bfloat16, device="cuda")
    ok(L.dic_cast_bf16_hl(p(Wd), p(hi), p(lo), N * K, stream()), L)
    ws = torch.zeros(L.dic_lo_mean_bias_ws_bytes(K) // 4, device="cuda")
    outs = []
    for _ in range(2):
        out = torch.full((N + 4,), 7.0, device="cuda")
        ok(L.dic_lo_mean_bias(p(Ad), T, K, stride, K, p(lo), K, N, p(dev(bias)), p(out), p(ws), stream()), L)
        torch.cuda.synchronize()
        outs.append(out.clone())
    assert torch.equal(outs[0], outs[1]) and bool((outs[0][N:] == 7.0).all())
    abar = Ad.float().cpu().double()[::stride].mean(0)
    ref = bias.double() + lo.float().cpu().double() @ abar
    corr = (ref - bias.double()).abs().max()
    assert float((outs[0][:N].cpu().double() - ref).abs().max()) < 1e-4 * float(corr) + 1e-7
    out0 = torch.zeros(N, device="cuda")                                                      # no bias given: the correction alone
    ok(L.dic_lo_mean_bias(p(Ad), T, K, stride, K, p(lo), K, N, 0, p(out0), p(ws), stream()), L)
    assert float((out0.cpu().double() - (ref - bias.double())).abs().max()) < 1e-4 * float(corr) + 1e-7


@pytest.mark.parametrize("T,K,N,stride,resid", [(17408, 768, 2304, 17, False), (17408, 768, 768, 17, True), (17408, 3072, 768, 17, True), (300, 768, 772, 1, False),
                                                (1000, 3072, 3072, 7, False), (333, 768, 768, 3, True)])
def test_lin_prep_mean_row_correction_and_reference_rows(L, T, K, N, stride, resid):
    """dic_lin_prep (round 5): bias_in = bias + W_lo . abar; for a residual Linear also y_ref = bias + (W_hi + W_lo) . abar + r_ref and
    bias_post = r_ref - y_ref (folded into bias_in when no dropout sits between).  abar = mean of every stride-th row.  Against float64 on the
    same bf16 operands; repeated launches on ONE workspace give the same bits, also right behind one another on the stream."""
    g = torch.Generator().manual_seed(T + K + N)
    A = torch.randn(T, K, generator=g) * 0.7 + torch.randn(1, K, generator=g) * 0.4
    W = torch.randn(N, K, generator=g) * 0.03
    bias, r_ref = torch.randn(N, generator=g) * 0.1, torch.randn(N, generator=g)
    Ad, Wd = dev(A, torch.bfloat16), dev(W)
    hi, lo = torch.zeros(N, K, dtype=torch.bfloat16, device="cuda"), torch.zeros(N, K, dtype=torch.bfloat16, device="cuda")
    ok(L.dic_cast_bf16_hl(p(Wd), p(hi), p(lo), N * K, stream()), L)
    ws = torch.zeros(L.dic_lin_prep_ws_bytes(K) // 4, device="cuda")
    abar = Ad.float().cpu().double()[::stride].mean(0)
    s_lo, s_hi = lo.float().cpu().double() @ abar, hi.float().cpu().double() @ abar
    outs = []
    for fold in ((0, 1) if resid else (0,)):
        for rep in range(3):
            b_in, b_post, y_ref = (torch.full((N + 4,), 7.0, device="cuda") for _ in range(3))
            ok(L.dic_lin_prep(p(Ad), T, K, stride, K, p(hi) if resid else 0, p(lo), K, N, p(dev(bias)), p(dev(r_ref)) if resid else 0, fold,
                              p(b_in), p(b_post) if resid else 0, p(y_ref) if resid else 0, p(ws), stream()), L)
            if rep == 0:
                torch.cuda.synchronize()
            outs.append((fold, b_in.clone(), b_post.clone(), y_ref.clone()))
    torch.cuda.synchronize()
    scale = float(s_lo.abs().max()) + 1e-9
    for fold, b_in, b_post, y_ref in outs:
        assert bool((b_in[N:] == 7.0).all())
        if resid:
            yr = bias.double() + s_lo + s_hi + r_ref.double()
            assert float((y_ref[:N].cpu().double() - yr).abs().max()) < 3e-6 * float(yr.abs().max())
            assert float((b_post[:N].cpu().double() - (r_ref.double() - yr)).abs().max()) < 3e-6 * float(yr.abs().max())
            want = bias.double() + s_lo + ((r_ref.double() - yr) if fold else 0)
            assert float((b_in[:N].cpu().double() - want).abs().max()) < 3e-6 * float(yr.abs().max()) + 1e-4 * scale
        else:
            assert float((b_in[:N].cpu().double() - (bias.double() + s_lo)).abs().max()) < 1e-4 * scale + 1e-7
    for a, b in zip(outs[:3], outs[1:3]):                                     # bit-identical reruns
        assert all(torch.equal(x, y) for x, y in zip(a[1:], b[1:]))


def test_gemm_second_bias_row_behind_the_dropout(L):
    """DicGemmParams.bias2: C = dropout(acc + bias) + bias2 + R (bf16 forward AFFINE with dropout and residual) -- same mask as without it."""
    M, N, K = 600, 768, 320
    g = torch.Generator().manual_seed(11)
    A, B = torch.randn(M, K, generator=g) * 0.3, torch.randn(N, K, generator=g) * 0.3
    bias, b2, Rr = torch.randn(N, generator=g), torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    Ad, Bd, Rd = dev(A, torch.bfloat16), dev(B, torch.bfloat16), dev(Rr, torch.bfloat16)
    for tile in (128, 256):
        C0 = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        C1 = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        kw = dict(A=p(Ad), B=p(Bd), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=p(dev(bias)), R=p(Rd), ldr=N, tile=tile, p_drop=0.2, seed=77)
        gemm(L, BF16, 0, 0, 0, C=p(C0), **kw)
        gemm(L, BF16, 0, 0, 0, C=p(C1), bias2=p(dev(b2)), **kw)
        want = (C0.float().cpu().double() + b2.double())
        assert float((C1.float().cpu().double() - want).abs().max()) < 2 ** -7 * float(want.abs().max())       # two bf16 roundings apart at most
    g_ = dic._lib.GemmParams()
    for k, v in dict(kw, C=p(C1), bias2=p(dev(b2)), p_drop=0.0).items():
        setattr(g_, k, v)
    assert L.dic_gemm(BF16, 0, 0, 0, C.byref(g_), stream()) != 0            # without dropout the row belongs into `bias`: refused loudly


def test_centred_layernorm_forward_backward(L):
    """dic_ln_fwd_cen / dic_ln_bwd_cen: LayerNorm of y = y_c + y_ref given as a bf16 remainder and one fp32 reference row; outputs the bf16 operand
    copy, the remainder of the output against h_ref = LN(y_ref), and h_ref.  With nearly equal rows (a collapsed denoiser) the centred pair
    carries the row differences that plain bf16 storage of y would round away."""
    T = 333
    g = torch.Generator().manual_seed(3)
    y_ref = torch.randn(768, generator=g) * 1.5 + 0.2
    gamma, beta = 1 + 0.1 * torch.randn(768, generator=g), 0.1 * torch.randn(768, generator=g)
    dh = torch.randn(T, 768, generator=g)
    for spread in (1.0, 1e-3):
        y = y_ref + spread * torch.randn(T, 768, generator=g)
        y_c = dev(y - y_ref, torch.bfloat16)
        y_used = (y_c.float().cpu() + y_ref).double().requires_grad_(True)
        gg, bb = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
        ref = R.layer_norm(y_used, gg, bb)
        # h_ref = LN of the reference row at the scale of a typical row: variance = mean over the four sample rows 0, T/4, T/2, 3T/4 (never below its own)
        yu = y_used.detach()
        v_typ = torch.stack([yu[(T * q) // 4].var(unbiased=False) for q in range(4)]).mean()
        yr64 = y_ref.double()
        rs = min(float(1.0 / torch.sqrt(v_typ + 1e-12)), float(1.0 / torch.sqrt(yr64.var(unbiased=False) + 1e-12)))
        href = (yr64 - yr64.mean()) * rs * gamma.double() + beta.double()
        h, h_c = (torch.zeros(T, 768, dtype=torch.bfloat16, device="cuda") for _ in range(2))
        h_ref, mean, rstd = torch.zeros(768, device="cuda"), torch.zeros(T, device="cuda"), torch.zeros(T, device="cuda")
        # (with the "next Linear" tail: bias_out = bias + W_lo . h_ref, the predicted mean-row correction of the Linear that reads h)
        g2 = torch.Generator().manual_seed(5)
        nlo, nb = dev(torch.randn(2304 + 5, 768, generator=g2) * 1e-3, torch.bfloat16), torch.randn(2304 + 5, generator=g2)
        nout = torch.full((2304 + 5 + 3,), 7.0, device="cuda")
        nhi = dev(torch.randn(2304 + 5, 768, generator=g2) * 3e-2, torch.bfloat16)
        ok(L.dic_ln_fwd_cen(p(y_c), p(dev(y_ref)), p(dev(gamma)), p(dev(beta)), p(h), p(h_c), p(h_ref), p(mean), p(rstd), T, 768, 1e-12,
                            0, p(nlo), 768, 2304 + 5, p(dev(nb)), p(nout), stream()), L)
        torch.cuda.synchronize()
        assert relerr(h_ref, href) < 2e-6
        want_nb = nb.double() + nlo.float().cpu().double() @ href
        assert float((nout[:2304 + 5].cpu().double() - want_nb).abs().max()) < 1e-5 and bool((nout[2304 + 5:] == 7.0).all())
        for use_lo in (True, False):                   # with the hi half: the whole reference-row term of a Linear that reads the CENTRED tensor
            ok(L.dic_ln_fwd_cen(p(y_c), p(dev(y_ref)), p(dev(gamma)), p(dev(beta)), 0, p(h_c), p(h_ref), p(mean), p(rstd), T, 768, 1e-12,
                                p(nhi), p(nlo) if use_lo else 0, 768, 2304 + 5, p(dev(nb)), p(nout), stream()), L)
            torch.cuda.synchronize()
            want2 = nb.double() + ((nlo.float().cpu().double() @ href) if use_lo else 0) + nhi.float().cpu().double() @ href
            assert float((nout[:2304 + 5].cpu().double() - want2).abs().max()) < 1e-5 * float(want2.abs().max()) + 1e-5
        assert relerr(h.float(), ref.detach()) < 5e-3
        rebuilt = h_c.float().cpu().double() + h_ref.cpu().double()
        dev_scale = float((ref.detach() - href).abs().max())
        assert float((rebuilt - ref.detach()).abs().max()) < 5e-3 * dev_scale + 1e-6          # error relative to the DISTANCE from the reference row
        h2 = torch.zeros_like(h)
        ok(L.dic_ln_fwd_cen(p(y_c), p(dev(y_ref)), p(dev(gamma)), p(dev(beta)), p(h2), 0, 0, p(mean), p(rstd), T, 768, 1e-12, 0, 0, 0, 0, 0, 0, stream()), L)
        torch.cuda.synchronize()
        assert torch.equal(h, h2)
        dhd = dev(dh, torch.bfloat16)
        ref.backward(dhd.float().cpu().double())
        dx, dxd = (torch.zeros(T, 768, dtype=torch.bfloat16, device="cuda") for _ in range(2))
        part = torch.zeros(64, 3 * 768, device="cuda")
        ok(L.dic_ln_bwd_cen(p(dhd), p(y_c), p(dev(y_ref)), p(dev(gamma)), p(mean), p(rstd), p(dx), 0, 0.0, 0, p(part), 64, T, 768, stream()), L)
        s = _colsum(L, part, 3 * 768)
        assert relerr(dx.float(), y_used.grad) < 1e-2
        assert relerr(s[:768], gg.grad) < 2e-5 and relerr(s[768:1536], bb.grad) < 1e-5


def test_rank_one_completion_of_a_centred_weight_gradient(L):
    """dic_rank1_add: dW += db x_ref^T -- together with the weight-gradient GEMM on the centred input X_c = X - 1 x_ref^T it gives dY^T X."""
    T, M, N = 64 * 9 + 3, 256, 264
    g = torch.Generator().manual_seed(4)
    dY, X, xr = torch.randn(T, M, generator=g), torch.randn(T, N, generator=g) * 0.05, torch.randn(N, generator=g)
    dYd, Xc = dev(dY, torch.bfloat16), dev(X, torch.bfloat16)
    dW, db = torch.zeros(M, N, device="cuda"), torch.zeros(M, device="cuda")
    item = dic._lib.WgradItem(dY=p(dYd), ldy=M, X=p(Xc), ldx=N, dW=p(dW), db=p(db), M=M, N=N)
    arr = (dic._lib.WgradItem * 1)(item)
    nbytes = L.dic_wgrad_group_ws_bytes(arr, 1, T, 0)
    ws = torch.empty(max(nbytes, 16) // 4, device="cuda")
    ok(L.dic_wgrad_group(arr, 1, T, p(ws), nbytes, 0, stream()), L)
    ok(L.dic_rank1_add(p(dW), p(db), p(dev(xr)), M, N, stream()), L)
    torch.cuda.synchronize()
    full = dYd.float().cpu().double().t() @ (Xc.float().cpu().double() + xr.double())
    assert relerr(dW, full) < 3e-6


def test_ln_and_gelu_ln_with_fp32_inputs_in_the_bf16_engine(L):
    """dic_ln_fwd_r32 / dic_ln_bwd(DIC_BF16 | DIC_RES_F32) and dic_gelu_ln_fwd / _bwd(DIC_BF16 | DIC_U_F32): fp32 y / u in, bf16 operand copy
    (+ fp32 residual copy) out, bf16 gradients -- the statistics and the fp32 outputs to fp32 accuracy, the bf16 outputs to bf16 rounding."""
    T = 333
    y, gamma, beta, dh = _ln_inputs(T, 5)
    yd, dhd = dev(y), dev(dh, torch.bfloat16)
    h = torch.zeros(T, 768, dtype=torch.bfloat16, device="cuda")
    h32 = torch.zeros(T, 768, device="cuda")
    mean, rstd = torch.zeros(T, device="cuda"), torch.zeros(T, device="cuda")
    ok(L.dic_ln_fwd_r32(p(yd), p(dev(gamma)), p(dev(beta)), p(h), p(h32), p(mean), p(rstd), T, 768, 1e-12, stream()), L)
    yy = y.double().requires_grad_(True)
    gg, bb = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = R.layer_norm(yy, gg, bb)
    assert relerr(h32, ref) < 2e-6 and torch.equal(h, h32.to(torch.bfloat16))
    h2 = torch.zeros_like(h)
    ok(L.dic_ln_fwd_r32(p(yd), p(dev(gamma)), p(dev(beta)), p(h2), 0, p(mean), p(rstd), T, 768, 1e-12, stream()), L)       # no fp32 copy asked for
    torch.cuda.synchronize()
    assert torch.equal(h, h2)
    ref.backward(dhd.float().cpu().double())
    dx = torch.zeros(T, 768, dtype=torch.bfloat16, device="cuda")
    part = torch.zeros(64, 3 * 768, device="cuda")
    ok(L.dic_ln_bwd(BF16 | 0x200, p(dhd), p(yd), p(dev(gamma)), p(mean), p(rstd), p(dx), 0, 0.0, 0, p(part), 64, T, 768, stream()), L)
    s = _colsum(L, part, 3 * 768)
    assert relerr(dx.float(), yy.grad) < 1e-2
    assert relerr(s[:768], gg.grad) < 1e-5 and relerr(s[768:1536], bb.grad) < 1e-5
    # GELU + LayerNorm of the MLM head with an fp32 pre-activation
    u, gamma, beta, dxo = _ln_inputs(T, 6)
    ud = dev(u)
    xo = torch.zeros(T, 768, device="cuda")
    ok(L.dic_gelu_ln_fwd(BF16 | 0x100, p(ud), p(dev(gamma)), p(dev(beta)), p(xo), p(mean), p(rstd), T, 768, 1e-12, stream()), L)
    uu = u.double().requires_grad_(True)
    gg, bb = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = R.layer_norm(R.gelu(uu), gg, bb)
    assert relerr(xo, ref) < 3e-6
    ref.backward(dxo.double())
    du = torch.zeros(T, 768, dtype=torch.bfloat16, device="cuda")
    part = torch.zeros(32, 3 * 768, device="cuda")
    ok(L.dic_gelu_ln_bwd(BF16 | 0x100, p(dev(dxo)), p(ud), p(dev(gamma)), p(mean), p(rstd), p(du), p(part), 32, T, 768, stream()), L)
    s = _colsum(L, part, 3 * 768)
    assert relerr(du.float(), uu.grad) < 1e-2
    assert relerr(s[:768], gg.grad) < 1e-5 and relerr(s[768:1536], bb.grad) < 1e-5


@pytest.mark.parametrize("split", [1, 3])
def test_gemm_fused_bias_gradient(L, split):
    """bf16 weight-gradient GEMM also returns colsum(dY) (the bias gradient) from the LDS-resident A tiles."""
    M, N, K = 384, 256, 64 * 11 + 5
    g = torch.Generator().manual_seed(40 + split)
    A, B = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
    Ad, Bd = dev(A, torch.bfloat16), dev(B, torch.bfloat16)
    Cd = torch.zeros(M, N, dtype=torch.float32, device="cuda")
    cs = torch.full((M,), float("nan"), device="cuda")
    ws = torch.zeros(split * (M * N + M), device="cuda")
    gemm(L, BF16, 1, 1, 0, A=p(Ad), B=p(Bd), C=p(Cd), M=M, N=N, K=K, lda=M, ldb=N, ldc=N, out_f32=1, split_k=split, split_ws=p(ws),
         colsum_out=p(cs))
    assert relerr(Cd, Ad.float().cpu().double().t() @ Bd.float().cpu().double()) < 2e-6
    assert relerr(cs, Ad.float().cpu().double().sum(0)) < 2e-6


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_gemm_f32_is_k_ordered_fmaf_chain_and_affine_epilogue(L, dtype):
    M, N, K = 136, 260, 128
    g = torch.Generator().manual_seed(5)
    A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    bias, Rr = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    Ad, Bd, Rd = dev(A, DT[dtype]), dev(B, DT[dtype]), dev(Rr, DT[dtype])
    Cd = torch.zeros(M, N, dtype=DT[dtype], device="cuda")
    gemm(L, dtype, 0, 0, 0, A=p(Ad), B=p(Bd), C=p(Cd), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=p(dev(bias)), R=p(Rd), ldr=N)
    ref = Ad.float().cpu().double() @ Bd.float().cpu().double().t() + bias.double() + Rd.float().cpu().double()
    assert relerr(Cd.float(), ref) < (1e-6 if dtype == F32 else 6e-3)
    if dtype == F32:   # bit-exact vs the fmaf-chain oracle (no bias / residual)
        Cd2 = torch.zeros(M, N, dtype=torch.float32, device="cuda")
        gemm(L, F32, 0, 0, 0, A=p(Ad), B=p(Bd), C=p(Cd2), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, out_f32=1)
        o = rounding_ref(A.numpy(), B.numpy(), None, want_logits=True)
        assert np.array_equal(Cd2.cpu().numpy(), o["logits"]), "fp32 MFMA GEMM is not the k-ascending fmaf chain"
        # accumulate=1
        gemm(L, F32, 0, 0, 0, A=p(Ad), B=p(Bd), C=p(Cd2), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, out_f32=1, accumulate=1)
        assert relerr(Cd2, 2 * torch.from_numpy(o["logits"])) < 1e-6


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_gemm_gelu_epilogues_and_dropout(L, dtype):
    M, N, K = 144, 256, 192
    g = torch.Generator().manual_seed(9)
    A, B, bias = torch.randn(M, K, generator=g) * 0.3, torch.randn(N, K, generator=g) * 0.3, torch.randn(N, generator=g)
    Ad, Bd = dev(A, DT[dtype]), dev(B, DT[dtype])
    U = torch.zeros(M, N, dtype=DT[dtype], device="cuda")
    G = torch.zeros(M, N, dtype=DT[dtype], device="cuda")
    gemm(L, dtype, 0, 0, 1, A=p(Ad), B=p(Bd), C=p(G), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=p(dev(bias)), aux=p(U), ldaux=N)
    u_ref = Ad.float().cpu().double() @ Bd.float().cpu().double().t() + bias.double()
    tol = 1e-5 if dtype == F32 else 1e-2
    assert relerr(U.float(), u_ref) < tol
    assert relerr(G.float(), R.gelu(U.float().cpu().double())) < tol
    # GELU backward epilogue: C = acc * gelu'(U)
    Dd = torch.zeros(M, N, dtype=DT[dtype], device="cuda")
    gemm(L, dtype, 0, 0, 2, A=p(Ad), B=p(Bd), C=p(Dd), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, aux=p(U), ldaux=N)
    uu = U.float().cpu().double().requires_grad_(True)
    R.gelu(uu).sum().backward()
    acc = Ad.float().cpu().double() @ Bd.float().cpu().double().t()
    assert relerr(Dd.float(), acc * uu.grad) < tol
    # dropout epilogue: kept elements scaled by 1/(1-p), drop rate ~ p, deterministic in (seed, index)
    C1 = torch.zeros(M, N, dtype=torch.float32, device="cuda")
    C2 = torch.zeros(M, N, dtype=torch.float32, device="cuda")
    for Cx in (C1, C2):
        gemm(L, dtype, 0, 0, 0, A=p(Ad), B=p(Bd), C=p(Cx), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, out_f32=1, p_drop=0.25, seed=1234)
    assert torch.equal(C1, C2)
    full = torch.zeros(M, N, dtype=torch.float32, device="cuda")
    gemm(L, dtype, 0, 0, 0, A=p(Ad), B=p(Bd), C=p(full), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, out_f32=1)
    kept = C1 != 0
    assert abs(float((~kept).float().mean()) - 0.25) < 0.02
    assert relerr(C1[kept], full[kept] / 0.75) < 1e-5


@pytest.mark.parametrize("tile,M,N,K,b_km", [(128, 144, 256, 192, 0), (128, 300, 776, 128, 1), (256, 600, 768, 320, 1), (256, 2176, 3072, 768, 1), (256, 520, 512, 64, 0)])
def test_gemm_gelu_derivative_epilogue_pair(L, tile, M, N, K, b_km):
    """BIAS_GELU_D / MUL_AUX (hf:221-223 and its backward): the forward leaves gelu'(u) behind instead of u, the backward's epilogue is one
    multiply.  g must equal BIAS_GELU's bit for bit (same arithmetic), the stored derivative is the fp64 derivative of the UNROUNDED u to bf16
    accuracy, and acc * aux matches the erf-form backward (GELU_BWD on the bf16 pre-activation) within bf16 rounding."""
    g = torch.Generator().manual_seed(9 + M)
    A, W, bias = torch.randn(M, K, generator=g) * 0.3, torch.randn(N, K, generator=g) * 0.3, torch.randn(N, generator=g)
    Ad, Wd = dev(A, torch.bfloat16), dev(W, torch.bfloat16)
    U, G1 = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda"), torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    D, G2 = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda"), torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    kw = dict(A=p(Ad), B=p(Wd), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=p(dev(bias)), ldaux=N, tile=tile)
    gemm(L, BF16, 0, 0, 1, C=p(G1), aux=p(U), **kw)
    gemm(L, BF16, 0, 0, 6, C=p(G2), aux=p(D), **kw)
    assert torch.equal(G1, G2)
    u64 = (Ad.float().cpu().double() @ Wd.float().cpu().double().t() + bias.double()).requires_grad_(True)
    R.gelu(u64).sum().backward()
    assert relerr(D.float(), u64.grad) < 6e-3
    G3 = torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda")
    gemm(L, BF16, 0, 0, 6, C=p(G3), aux=0, **kw)                       # forward-only call: nothing kept
    assert torch.equal(G3, G1)
    # backward: dU = dY W2-shaped product * aux; operands of the engine's call: A = dY [M][Kb], B = W2 stored k-major [Kb][N]
    Kb = 192
    dY, W2 = torch.randn(M, Kb, generator=g) * 0.3, torch.randn(N, Kb, generator=g) * 0.3
    dYd = dev(dY, torch.bfloat16)
    W2d = dev(W2.t() if b_km else W2, torch.bfloat16)
    acc = dYd.float().cpu().double() @ (W2d.float().cpu().double() if b_km else W2d.float().cpu().double().t())
    O1, O2 = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda"), torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    kb = dict(A=p(dYd), B=p(W2d), M=M, N=N, K=Kb, lda=Kb, ldb=(N if b_km else Kb), ldc=N, ldaux=N, tile=tile)
    gemm(L, BF16, 0, b_km, 7, C=p(O1), aux=p(D), **kb)
    assert relerr(O1.float(), acc * D.float().cpu().double()) < 6e-3
    gemm(L, BF16, 0, b_km, 2, C=p(O2), aux=p(U), **kb)
    assert relerr(O1.float(), O2.float()) < 2e-2


@pytest.mark.parametrize("M,N,K,epi", [(17408, 2304, 768, 0), (17408, 3072, 768, 1), (34816, 768, 768, 0), (34816, 3072, 768, 1), (29920, 768, 3072, 0),
                                       (38080, 768, 768, 0), (20000, 1024, 192, 0)])
def test_gemm_two_tile_heights_in_one_launch(L, M, N, K, epi):
    """A forward GEMM whose units would end in a partly filled round of the persistent grid runs whole rounds of tall tiles + one round of
    shorter tiles (gemm_bf16_kernel2): bit-identical to the one-height launch -- output, GELU pre-activation, and the dropout mask, which is
    keyed by the global row -- and equal to torch on the plain product.  The bench / sampling shapes must actually take the two-height path."""
    plan = (C.c_int * 3)()
    ok(L.dic_gemm_two_heights_plan(M, N, K, 0, plan), L)
    if (M, N, K) in ((17408, 2304, 768), (17408, 3072, 768), (34816, 768, 768), (34816, 3072, 768)):
        assert plan[0] in (7, 8) and 4 <= plan[1] <= 7 and 0 < plan[2] < M and plan[2] % (32 * plan[0]) == 0, list(plan)
    g = torch.Generator().manual_seed(M + N)
    A = dev(torch.randn(M, K, generator=g), DT[BF16]); B = dev(torch.randn(N, K, generator=g) * 0.05, DT[BF16])
    bias = dev(torch.randn(N, generator=g))
    R = dev(torch.randn(M, N, generator=g), DT[BF16]) if epi == 0 else None
    outs = []
    for on in (0, 1):
        prev = L.dic_gemm_set_two_heights(on)
        Cc = torch.full((M, N), float("nan"), dtype=DT[BF16], device="cuda")
        U = torch.full((M, N), float("nan"), dtype=DT[BF16], device="cuda") if epi == 1 else None
        kw = dict(A=p(A), B=p(B), C=p(Cc), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=p(bias), tile=256)
        if epi == 0:
            kw.update(R=p(R), ldr=N, p_drop=0.1, seed=77)
        else:
            kw.update(aux=p(U), ldaux=N)
        gemm(L, BF16, 0, 0, epi, **kw)
        outs.append((Cc, U))
        L.dic_gemm_set_two_heights(prev)
    assert torch.equal(outs[0][0], outs[1][0])
    if epi == 1:
        assert torch.equal(outs[0][1], outs[1][1])
        ref = A[:4096].float() @ B.float().t() + bias
        assert relerr(outs[1][1][:4096].float(), ref) < 1e-2
        tail = A[-2048:].float() @ B.float().t() + bias
        assert relerr(outs[1][1][-2048:].float(), tail) < 1e-2


# ------------------------------------------------------------------------------------------------ rounding head
@pytest.mark.parametrize("dtype,V,tile", [(F32, 30522, 128), (BF16, 30522, 128), (F32, 1000, 128), (BF16, 30522, 256), (BF16, 1000, 256)])
def test_rounding_ce_partial_combine_and_backward(L, dtype, V, tile):
    M, K = (40 if tile == 128 else 300), 768
    g = torch.Generator().manual_seed(V + dtype)
    x = torch.randn(M, K, generator=g)
    W = torch.randn(V, K, generator=g) * 0.05
    tgt = torch.randint(0, V, (M,), generator=g)
    # plant exact ties: duplicate rows of W -> equal logits; first index must win
    W[7] = W[3]
    W[V - 1] = W[V - 2]
    vpad = (V + 127) // 128 * 128
    Wp = torch.zeros(vpad, K)
    Wp[:V] = W
    xd, Wd, td = dev(x, DT[dtype]), dev(Wp, DT[dtype]), dev(tgt)
    npart = 2 * ((V + 127) // 128) if tile == 128 else 4 * ((V + 255) // 256)      # one record per 64 columns of every tile
    part = torch.zeros(M, npart, 4, device="cuda")
    tl = torch.zeros(M, device="cuda")
    lse = torch.zeros(M, device="cuda")
    am = torch.zeros(M, dtype=torch.int64, device="cuda")
    nll = torch.zeros(M, device="cuda")
    gemm(L, dtype, 0, 0, 3, A=p(xd), B=p(Wd), M=M, N=V, K=K, lda=K, ldb=K, tgt=p(td), partial=p(part), tgt_logit=p(tl), tile=tile)
    ok(L.dic_ce_combine(p(part), p(tl), M, npart, p(lse), p(am), p(nll), stream()), L)
    torch.cuda.synchronize()
    o = rounding_ref(xd.float().cpu().numpy(), Wd.float().cpu().numpy()[:V], tgt.numpy())
    if dtype == F32:
        assert np.array_equal(am.cpu().numpy(), o["argmax"]), "token ids must be bit-exact (first index on ties)"
        assert np.array_equal(tl.cpu().numpy(), o["tgt_logit"])
    else:
        logits = xd.float().cpu().double() @ Wd.float().cpu().double()[:V].t()
        top = logits.max(-1).values
        chosen = logits.gather(1, am.cpu().unsqueeze(1)).squeeze(1)
        assert float((top - chosen).max()) < 1e-3
    # (the oracle sees the SAME bf16-rounded operands: what is tested is the streaming logsumexp / gather itself, measured at 3e-8 relative on
    # 16 384 rows, profiles/r03_ce_gap_probe.txt line A; the bf16 engine's distance from the fp32 engine comes from its activations, not from here)
    np.testing.assert_allclose(lse.cpu().numpy(), o["lse"], rtol=2e-6 if dtype == F32 else 2e-5)
    np.testing.assert_allclose(nll.cpu().numpy(), o["lse"] - o["tgt_logit"], rtol=1e-5 if dtype == F32 else 1e-4, atol=1e-5)
    # backward: dlogits epilogue + (KC,KM) GEMM against autograd of the oracle's rounding loss
    rows_a, sa, sb = M * 3 // 5, 0.5 / 3, 0.5 / 2
    dlog = torch.full((M, vpad), float("nan"), dtype=DT[dtype], device="cuda")
    dxr = torch.zeros(M, K, dtype=torch.float32, device="cuda")
    gemm(L, dtype, 0, 0, 4, A=p(xd), B=p(Wd), C=p(dlog), M=M, N=V, K=K, lda=K, ldb=K, ldc=vpad, tgt=p(td), lse=p(lse),
         ce_rows_a=rows_a, ce_scale_a=sa, ce_scale_b=sb)
    assert float(dlog[:, V:].float().abs().max()) == 0.0, "padding columns of dlogits must be zero"
    gemm(L, dtype, 0, 1, 0, A=p(dlog), B=p(Wd), C=p(dxr), M=M, N=K, K=vpad, lda=vpad, ldb=K, ldc=K, out_f32=1)
    xx = xd.float().cpu().double().requires_grad_(True)
    lg = xx @ Wd.float().cpu().double()[:V].t()
    nl = -torch.log_softmax(lg, -1).gather(1, tgt.unsqueeze(1)).squeeze(1)
    (nl[:rows_a].sum() * sa + nl[rows_a:].sum() * sb).backward()
    assert relerr(dxr, xx.grad) < (1e-5 if dtype == F32 else 2e-2)


@pytest.mark.parametrize("V,tile,n_a,n_b", [(30522, 256, 40, 24), (1000, 128, 9, 0), (30522, 256, 1024, 1024)])
def test_mean_centred_head_input_gives_the_same_logits_with_less_rounding(L, V, tile, n_a, n_b):
    """dic_head_center (ref:323 evaluated as (x - xbar) W^T + xbar W^T): xbar / cvec / xr against torch; then the eval (CE_PARTIAL) and the training
    (CE_EXP) form of the rounding loss on the centred input + bias against float64 CE on the UNROUNDED rows -- for rows that share a large common
    vector (what an early-training denoiser produces: |xbar| = 27, deviations 0.01) the plain bf16 head is 1e-4-class off in the batch mean while the
    centred one is at fp32 round-off, and for ordinary rows both agree."""
    K, Lh, Tk = 768, 4, 5
    g = torch.Generator().manual_seed(V + n_a)
    W = torch.randn(V, K, generator=g) * 0.05
    vpad = (V + 127) // 128 * 128
    Wp = torch.zeros(vpad, K)
    Wp[:V] = W
    common = torch.randn(K, generator=g)
    xa = common + 0.01 * torch.randn(n_a, Tk, K, generator=g)
    xb = common + 0.01 * torch.randn(max(n_b, 1), Tk, K, generator=g)
    M = (n_a + n_b) * Lh
    rows = torch.cat([xa[:, :Lh].reshape(-1, K), xb[:n_b, :Lh].reshape(-1, K)])
    tgt = torch.randint(0, V, (M,), generator=g)
    xad, xbd, W32, Wc, td = dev(xa), dev(xb), dev(Wp), dev(Wp, DT[BF16]), dev(tgt)
    ws = torch.empty(L.dic_head_center_ws_bytes(K) // 4, device="cuda")
    xbar, cvec = torch.zeros(K, device="cuda"), torch.zeros(vpad + 256, device="cuda")
    xr = torch.zeros(M, K, dtype=DT[BF16], device="cuda")
    ok(L.dic_head_center(p(xad), n_a, p(xbd) if n_b else 0, n_b, Lh, Tk, K, p(W32), vpad, p(ws), p(xbar), p(cvec), p(xr), stream()), L)
    torch.cuda.synchronize()
    mean = rows.double().mean(0)
    np.testing.assert_allclose(xbar.cpu().numpy(), mean.numpy(), rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(cvec[:V].cpu().numpy(), (W.double() @ xbar.cpu().double()).numpy(), rtol=1e-5, atol=1e-5)
    assert float(cvec[V:vpad].abs().max()) == 0.0
    assert torch.equal(xr.cpu(), (rows - xbar.cpu()).to(DT[BF16]))
    lg = rows.double() @ W.double().t()
    ref_nll = torch.logsumexp(lg, -1) - lg.gather(1, tgt.unsqueeze(1)).squeeze(1)
    npart = L.dic_ce_n_partials(V, tile)

    def eval_form(xin, bias):
        part = torch.zeros(M, npart, 4, device="cuda")
        tl, lse, nll = (torch.zeros(M, device="cuda") for _ in range(3))
        am = torch.zeros(M, dtype=torch.int64, device="cuda")
        gemm(L, BF16, 0, 0, 3, A=p(xin), B=p(Wc), C=0, M=M, N=V, K=K, lda=K, ldb=K, ldc=0, tgt=p(td), partial=p(part), tgt_logit=p(tl), tile=tile, bias=bias)
        ok(L.dic_ce_combine(p(part), p(tl), M, npart, p(lse), p(am), p(nll), stream()), L)
        torch.cuda.synchronize()
        return nll.cpu().double(), am.cpu()

    def train_form(xin, bias):
        part = torch.zeros(M, npart, device="cuda")
        t0, cref, tl, lse, nll, inv_z = (torch.zeros(M, device="cuda") for _ in range(6))
        E = torch.zeros(M, vpad, dtype=DT[BF16], device="cuda")
        ok(L.dic_ce_target_logit(p(xin), p(Wc), p(td), M, V, K, 40.0, p(t0), p(cref), bias, stream()), L)
        gemm(L, BF16, 0, 0, 5, A=p(xin), B=p(Wc), C=p(E), M=M, N=V, K=K, lda=K, ldb=K, ldc=vpad, tgt=p(td), lse=p(cref), partial=p(part), tgt_logit=p(tl), tile=tile, bias=bias)
        ok(L.dic_ce_exp_combine(p(part), npart, p(cref), p(tl), p(td), M, V, p(E), vpad, p(lse), p(nll), p(inv_z), stream()), L)
        torch.cuda.synchronize()
        return nll.cpu().double()
    plain_in = dev(rows, DT[BF16])
    nll_plain, _ = eval_form(plain_in, 0)
    nll_cent, am = eval_form(xr, p(cvec))
    nll_cent_t = train_form(xr, p(cvec))
    e_plain = abs(float(nll_plain.mean() - ref_nll.mean())) / float(ref_nll.mean())
    e_cent = abs(float(nll_cent.mean() - ref_nll.mean())) / float(ref_nll.mean())
    e_cent_t = abs(float(nll_cent_t.mean() - ref_nll.mean())) / float(ref_nll.mean())
    print(f"batch-mean nll vs float64 on the unrounded rows: plain bf16 head {e_plain:.2e}, centred eval form {e_cent:.2e}, centred training form {e_cent_t:.2e}")
    assert e_cent < 2e-6 and e_cent_t < 2e-6 and e_cent < e_plain
    np.testing.assert_allclose(nll_cent.numpy(), ref_nll.numpy(), rtol=2e-4, atol=2e-4)
    assert float((am == lg.argmax(-1)).float().mean()) > 0.9          # (near-ties between 30 522 logits of almost identical rows may flip)


@pytest.mark.parametrize("V,tile,M", [(30522, 256, 300), (30522, 128, 40), (1000, 256, 300), (1000, 128, 75)])
def test_rounding_training_form_exp_epilogue_equals_softmax_minus_onehot(L, V, tile, M):
    """dic_ce_target_logit -> dic_gemm(CE_EXP) -> dic_ce_exp_combine (bf16): lse / nll as the streaming form gives them, inv_z * E equal to
    softmax - onehot per element to bf16 rounding (what CE_DLOGITS writes after a second GEMM), padding columns zero, and the gradient
    inv_z * row_scale * (E @ W) against autograd of the oracle's rounding loss on the same bf16 operands.  One row's logits are shifted far
    from its target's (a confidently wrong row) and one target is made dominant: the reference point keeps both finite."""
    K = 768
    g = torch.Generator().manual_seed(V + tile)
    x = torch.randn(M, K, generator=g)
    W = torch.randn(V, K, generator=g) * 0.05
    tgt = torch.randint(0, V, (M,), generator=g)
    x[1] = 18.0 * W[(int(tgt[1]) + 5) % V] / W[(int(tgt[1]) + 5) % V].norm() * 4        # another token wins by a wide margin (nll ~ 50)
    x[2] = 18.0 * W[int(tgt[2])] / W[int(tgt[2])].norm() * 4                            # the target wins by a wide margin (nll ~ 0)
    vpad = (V + 127) // 128 * 128
    Wp = torch.zeros(vpad, K)
    Wp[:V] = W
    xd, Wd, td = dev(x, DT[BF16]), dev(Wp, DT[BF16]), dev(tgt)
    npart = L.dic_ce_n_partials(V, tile)
    part = torch.full((M, npart), float("nan"), device="cuda")
    t0, cref, tl = (torch.zeros(M, device="cuda") for _ in range(3))
    lse, nll, inv_z = (torch.zeros(M, device="cuda") for _ in range(3))
    E = torch.full((M, vpad), float("nan"), dtype=DT[BF16], device="cuda")
    ok(L.dic_ce_target_logit(p(xd), p(Wd), p(td), M, V, K, 40.0, p(t0), p(cref), 0, stream()), L)
    gemm(L, BF16, 0, 0, 5, A=p(xd), B=p(Wd), C=p(E), M=M, N=V, K=K, lda=K, ldb=K, ldc=vpad, tgt=p(td), lse=p(cref), partial=p(part), tgt_logit=p(tl), tile=tile)
    ok(L.dic_ce_exp_combine(p(part), npart, p(cref), p(tl), p(td), M, V, p(E), vpad, p(lse), p(nll), p(inv_z), stream()), L)
    torch.cuda.synchronize()
    xx = xd.float().cpu().double().requires_grad_(True)
    W64 = Wd.float().cpu().double()[:V]
    lg = xx @ W64.t()
    ref_lse = torch.logsumexp(lg, -1)
    ref_t = lg.gather(1, tgt.unsqueeze(1)).squeeze(1)
    assert float(ref_lse[1] - ref_t[1]) > 30 and float(ref_lse[2] - ref_t[2]) < 1e-3            # the two planted rows are what they claim
    np.testing.assert_allclose(t0.cpu().numpy(), ref_t.detach().numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(tl.cpu().numpy(), ref_t.detach().numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(lse.cpu().numpy(), ref_lse.detach().numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(nll.cpu().numpy(), (ref_lse - ref_t).detach().numpy(), rtol=1e-4, atol=3e-5)
    assert float(E[:, V:].float().abs().max()) == 0.0, "padding columns must be zero"
    dl = E[:, :V].float().cpu().double() * inv_z.cpu().double().unsqueeze(1)
    ref_dl = torch.softmax(lg.detach(), -1)
    ref_dl[torch.arange(M), tgt] -= 1.0
    err = (dl - ref_dl).abs()
    assert float((err / (ref_dl.abs() + 1e-6)).max()) < 6e-3, "every element within bf16 rounding (2^-8) of softmax - onehot"
    rows_a, sa, sb = M * 3 // 5, 0.5 / 3, 0.5 / 2
    dxr = torch.zeros(M, K, dtype=torch.float32, device="cuda")
    gemm(L, BF16, 0, 1, 0, A=p(E), B=p(Wd), C=p(dxr), M=M, N=K, K=vpad, lda=vpad, ldb=K, ldc=K, out_f32=1)
    dx = torch.zeros(M, 1, K, device="cuda")                     # [N = M sequences][Tk = 1][768], L = 1: add_rows' layout with one row each
    ok(L.dic_add_rows_scaled(p(dx), p(dxr), p(inv_z), sa, rows_a, 1, 1, K, stream()), L)
    ok(L.dic_add_rows_scaled(p(dx) + rows_a * K * 4, p(dxr) + rows_a * K * 4, p(inv_z) + rows_a * 4, sb, M - rows_a, 1, 1, K, stream()), L)
    nl = ref_lse - ref_t
    (nl[:rows_a].sum() * sa + nl[rows_a:].sum() * sb).backward()
    assert relerr(dx.reshape(M, K), xx.grad) < 2e-2
    # saturation instead of overflow: a row whose winning logit is ~190 nats above its target's still gives finite statistics and a finite gradient
    x2 = x.clone()
    x2[1] = x[1] * 1.9
    xd2 = dev(x2, DT[BF16])
    ok(L.dic_ce_target_logit(p(xd2), p(Wd), p(td), M, V, K, 40.0, p(t0), p(cref), 0, stream()), L)
    gemm(L, BF16, 0, 0, 5, A=p(xd2), B=p(Wd), C=p(E), M=M, N=V, K=K, lda=K, ldb=K, ldc=vpad, tgt=p(td), lse=p(cref), partial=p(part), tgt_logit=p(tl), tile=tile)
    ok(L.dic_ce_exp_combine(p(part), npart, p(cref), p(tl), p(td), M, V, p(E), vpad, p(lse), p(nll), p(inv_z), stream()), L)
    gemm(L, BF16, 0, 1, 0, A=p(E), B=p(Wd), C=p(dxr), M=M, N=K, K=vpad, lda=vpad, ldb=K, ldc=K, out_f32=1)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(nll).all()) and bool(torch.isfinite(dxr * inv_z.unsqueeze(1)).all()) and 100 < float(nll[1]) < 115
    lg2 = xd2.float().cpu().double() @ W64.t()
    keep = torch.ones(M, dtype=torch.bool); keep[1] = False
    np.testing.assert_allclose(nll.cpu().numpy()[keep.numpy()], (torch.logsumexp(lg2, -1) - lg2.gather(1, tgt.unsqueeze(1)).squeeze(1)).numpy()[keep.numpy()], rtol=1e-4, atol=3e-5)


# ------------------------------------------------------------------------------------------------ embedding + q_sample
def test_embed_gather_reports_out_of_range_ids(L):
    """nn.Embedding raises for ids outside [0, V); the kernel reports them (count + last position) and zero-fills the row -- no clamping."""
    V = 500
    E = torch.randn(V, 768, generator=torch.Generator().manual_seed(1))
    ids = torch.tensor([[3, V, 7, -1], [V - 1, 0, 12345, 9]])
    out = torch.full((2, 4, 768), float("nan"), device="cuda")
    err = torch.zeros(2, dtype=torch.int32, device="cuda")
    ok(L.dic_embed_gather(p(dev(ids)), p(dev(E)), p(out), 8, 768, V, p(err), stream()), L)
    torch.cuda.synchronize()
    assert err.tolist() == [3, 7]                                 # three bad ids, the last one at flat position 6
    good = (ids >= 0) & (ids < V)
    assert torch.equal(out.cpu()[good], E[ids[good]]) and bool((out.cpu()[~good] == 0).all())
    # through the drop-in: model.embedding() surfaces it as IndexError (at the next lookup, or at once with check_ids(sync=True))
    dic.cfg.update(VOCAB_SIZE=V, TRAIN_EMBEDDING=False, IN_CHANNEL=768)
    model = dic.DistilBertModel(E.numpy(), E.numpy(), config=dict(n_layers=1, dropout=0.0, attention_dropout=0.0), dtype="fp32")
    model.embedding(ids.clamp(0, V - 1))
    model.check_ids(sync=True)                                    # clean batch: nothing raised
    model.embedding(ids)
    with pytest.raises(IndexError):
        model.check_ids(sync=True)
    model.embedding(ids)
    torch.cuda.synchronize()
    with pytest.raises(IndexError):
        model.embedding(ids.clamp(0, V - 1))                      # the next lookup reports the previous batch


def test_embed_gather_and_qsample_bit_exact(L):
    V, B, Lq, S = 500, 3, 16, 5
    g = torch.Generator().manual_seed(3)
    E = torch.randn(V, 768, generator=g)
    ids = torch.randint(0, V, (B, Lq), generator=g)
    out = torch.zeros(B, Lq, 768, device="cuda")
    ok(L.dic_embed_gather(p(dev(ids)), p(dev(E)), p(out), B * Lq, 768, V, 0, stream()), L)
    assert torch.equal(out.cpu(), E[ids])
    for cosine, T in ((True, 1000), (False, 100)):
        cfg = R.Config(COSIN_SCHEDULE=cosine, STEP_TOT=T)
        ac = R.alpha_cumprod(cfg)
        t = torch.randint(0, T, (S, 1, 1), generator=g)
        noise = torch.randn(B, Lq, 768, generator=g)
        x0 = E[ids]
        ref = R.diffuse_t(x0, t, noise, ac)
        xt = torch.zeros(S * B, Lq, 768, device="cuda")
        ok(L.dic_qsample(p(dev(x0)), p(dev(noise)), p(dev(t.reshape(-1))), p(dev(torch.sqrt(ac))), p(dev(torch.sqrt(1 - ac))), p(xt), 0, S, B,
                         Lq * 768, T, 0, stream()), L)
        assert torch.equal(xt.cpu(), ref), "q_sample must be bit-exact with the reference arithmetic"
    # device RNG: N(0,1) moments, one draw shared by all S, reproducible per seed
    nz = torch.zeros(B, Lq, 768, device="cuda")
    x0z = torch.zeros(64, 16, 768, device="cuda")
    big = torch.zeros(2 * 64, 16, 768, device="cuda")
    nzb = torch.zeros(64, 16, 768, device="cuda")
    tt = dev(torch.tensor([50, 50]))
    acd = dev(R.alpha_cumprod(R.Config(COSIN_SCHEDULE=False, STEP_TOT=100)))
    ok(L.dic_qsample(p(x0z), 0, p(tt), p(dev(torch.sqrt(acd))), p(dev(torch.sqrt(1 - acd))), p(big), p(nzb), 2, 64, 16 * 768, 100, 42, stream()), L)
    torch.cuda.synchronize()
    e = nzb.cpu()
    assert abs(float(e.mean())) < 5e-3 and abs(float(e.var()) - 1.0) < 1e-2 and abs(float((e ** 4).mean()) - 3.0) < 0.1
    assert torch.equal(big[:64], big[64:])


# ------------------------------------------------------------------------------------------------ LayerNorm family
def _ln_inputs(T, seed):
    g = torch.Generator().manual_seed(seed)
    y = torch.randn(T, 768, generator=g) * 1.7 + 0.3
    gamma = 1 + 0.1 * torch.randn(768, generator=g)
    beta = 0.1 * torch.randn(768, generator=g)
    dh = torch.randn(T, 768, generator=g)
    return y, gamma, beta, dh


def _colsum(L, part, cols):
    out = torch.zeros(cols, device="cuda")
    ws = torch.zeros(64 * cols, device="cuda")
    ok(L.dic_colsum(F32, p(part), part.shape[0], cols, cols, p(out), 0, p(ws), stream()), L)
    torch.cuda.synchronize()
    return out.cpu()


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_ln_fwd_bwd(L, dtype):
    T = 333
    y, gamma, beta, dh = _ln_inputs(T, 1)
    yd, dhd = dev(y, DT[dtype]), dev(dh, DT[dtype])
    h = torch.zeros(T, 768, dtype=DT[dtype], device="cuda")
    mean, rstd = torch.zeros(T, device="cuda"), torch.zeros(T, device="cuda")
    ok(L.dic_ln_fwd(dtype, p(yd), p(dev(gamma)), p(dev(beta)), p(h), p(mean), p(rstd), T, 768, 1e-12, stream()), L)
    yy = yd.float().cpu().double().requires_grad_(True)
    gg, bb = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = R.layer_norm(yy, gg, bb)
    tol = 2e-6 if dtype == F32 else 8e-3
    assert relerr(h.float(), ref) < tol
    ref.backward(dhd.float().cpu().double())
    dx = torch.zeros(T, 768, dtype=DT[dtype], device="cuda")
    part = torch.zeros(64, 3 * 768, device="cuda")
    ok(L.dic_ln_bwd(dtype, p(dhd), p(yd), p(dev(gamma)), p(mean), p(rstd), p(dx), 0, 0.0, 0, p(part), 64, T, 768, stream()), L)
    s = _colsum(L, part, 3 * 768)
    assert relerr(dx.float(), yy.grad) < (1e-5 if dtype == F32 else 1e-2)
    assert relerr(s[:768], gg.grad) < 1e-5 and relerr(s[768:1536], bb.grad) < 1e-5
    assert relerr(s[1536:], dx.float().cpu().double().sum(0)) < (1e-4 if dtype == F32 else 5e-3)   # kernel sums before bf16 rounding


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_gelu_ln_fwd_bwd(L, dtype):
    T = 150
    u, gamma, beta, dxo = _ln_inputs(T, 2)
    ud = dev(u, DT[dtype])
    xo = torch.zeros(T, 768, device="cuda")
    mean, rstd = torch.zeros(T, device="cuda"), torch.zeros(T, device="cuda")
    ok(L.dic_gelu_ln_fwd(dtype, p(ud), p(dev(gamma)), p(dev(beta)), p(xo), p(mean), p(rstd), T, 768, 1e-12, stream()), L)
    uu = ud.float().cpu().double().requires_grad_(True)
    gg, bb = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = R.layer_norm(R.gelu(uu), gg, bb)
    assert relerr(xo, ref) < 3e-6
    ref.backward(dxo.double())
    du = torch.zeros(T, 768, dtype=DT[dtype], device="cuda")
    part = torch.zeros(32, 3 * 768, device="cuda")
    ok(L.dic_gelu_ln_bwd(dtype, p(dev(dxo)), p(ud), p(dev(gamma)), p(mean), p(rstd), p(du), p(part), 32, T, 768, stream()), L)
    s = _colsum(L, part, 3 * 768)
    assert relerr(du.float(), uu.grad) < (1e-5 if dtype == F32 else 1e-2)
    assert relerr(s[:768], gg.grad) < 1e-5 and relerr(s[768:1536], bb.grad) < 1e-5


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("with_temb", [False, True])
def test_fuse_ln_fwd_bwd(L, dtype, mode, with_temb):
    """with_temb: the optional timestep-embedding operand (off = NULL = the reference's arithmetic, ref :271 takes no t)."""
    N, Lq = 5, 16
    Tk = Lq + 2 if mode == 0 else Lq
    g = torch.Generator().manual_seed(11 + mode)
    x = torch.randn(N, Lq, 768, generator=g)
    img, txt = torch.randn(N, 768, generator=g), torch.randn(N, 768, generator=g)
    seg, pos = torch.randn(2, 768, generator=g), 0.02 * torch.randn(512, 768, generator=g)
    gamma, beta = 1 + 0.1 * torch.randn(768, generator=g), 0.1 * torch.randn(768, generator=g)
    add_txt = torch.tensor([0, 1, 0, 1, 1], dtype=torch.uint8)
    dh = torch.randn(N, Tk, 768, generator=g)
    steps = 7
    temb = 0.5 * torch.randn(steps, 768, generator=g)
    tidx = torch.tensor([3, -1, 3, 0, 6], dtype=torch.int32)            # sequence 1 carries none; timestep 3 occurs twice
    leaves = [t.double().requires_grad_(True) for t in (img, txt, seg, pos, gamma, beta, temb)]
    im, tx, sg, ps, gm, bt, te = leaves
    if mode == 0:
        rows = torch.cat([x.double(), im[:, None], tx[:, None]], 1) + sg[torch.tensor([0] * Lq + [1] * 2)]
    else:
        rows = x.double() + im[:, None] + tx[:, None] * add_txt.double()[:, None, None]
    rows = rows + ps[:Tk]
    if with_temb:
        rows = rows + (te[tidx.clamp(min=0).long()] * (tidx >= 0).double()[:, None])[:, None, :]
    rows.retain_grad()
    ref = R.layer_norm(rows, gm, bt)
    h = torch.zeros(N, Tk, 768, dtype=DT[dtype], device="cuda")
    mean, rstd = torch.zeros(N * Tk, device="cuda"), torch.zeros(N * Tk, device="cuda")
    tidx_d = dev(tidx)
    args = (p(dev(x)), p(dev(img)), p(dev(txt)), p(dev(add_txt)), p(dev(seg)), p(dev(pos)), p(dev(temb)) if with_temb else 0,
            p(tidx_d) if with_temb else 0, p(dev(gamma)))
    ok(L.dic_fuse_ln_fwd(dtype, mode, *args, p(dev(beta)), p(h), p(mean), p(rstd), N, Lq, 768, 1e-12, 0.0, 0, stream()), L)
    assert relerr(h.float(), ref) < (3e-6 if dtype == F32 else 8e-3)
    dhd = dev(dh, DT[dtype])
    ref.backward(dhd.float().cpu().double())
    dy = torch.zeros(N, Tk, 768, device="cuda")
    part = torch.zeros(16, 2 * 768, device="cuda")
    ok(L.dic_fuse_ln_bwd(dtype, mode, *args, p(dhd), p(mean), p(rstd), p(dy), p(part), 16, N, Lq, 768, 0.0, 0, stream()), L)
    s = _colsum(L, part, 2 * 768)
    assert relerr(dy, rows.grad) < 1e-5
    assert relerr(s[:768], gm.grad) < 1e-5 and relerr(s[768:], bt.grad) < 1e-5
    if with_temb:
        dte = torch.full((steps, 768), float("nan"), device="cuda")
        ok(L.dic_temb_grad(p(dy), p(tidx_d), N, Tk, 768, steps, p(dte), stream()), L)
        assert relerr(dte, te.grad) < 1e-5 and bool((dte[[1, 2, 4, 5]] == 0).all())


# ------------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("dtype,Tk", [(F32, 18), (F32, 34), (F32, 16), (BF16, 18), (BF16, 17), (BF16, 16), (BF16, 32), (BF16, 33), (BF16, 34),
                                      (BF16, 47), (BF16, 63), (BF16, 64)])
def test_attention_fwd_bwd(L, dtype, Tk):
    """bf16: one 32-token MFMA tile up to 32 tokens, the 2 x 2-tile MFMA kernel for 33..64 (seq_len 32 + CLIP rows = 34); fp32: VALU kernel."""
    N, H, D = 3, 12, 768
    g = torch.Generator().manual_seed(Tk + dtype)
    qkv = torch.randn(N, Tk, 3 * D, generator=g)
    dctx = torch.randn(N, Tk, D, generator=g)
    km = torch.ones(N, Tk, dtype=torch.uint8)
    km[0, 5:Tk - 2] = 0
    km[1, Tk - 1] = 0
    km[2, 9:] = 0
    km[2, 0] = 1
    qd, dd = dev(qkv, DT[dtype]), dev(dctx, DT[dtype])
    ctx = torch.full((N, Tk, D), float("nan"), dtype=DT[dtype], device="cuda")
    ok(L.dic_attn_fwd(dtype, p(qd), p(dev(km)), p(ctx), N, Tk, H, 64, 0.0, 0, stream()), L)
    qq = qd.float().cpu().double().requires_grad_(True)
    ref = R.attention(qq[..., :D], qq[..., D:2 * D], qq[..., 2 * D:], km, H)
    tol = 1e-5 if dtype == F32 else 1.5e-2
    assert relerr(ctx.float(), ref) < tol
    ref.backward(dd.float().cpu().double())
    dq = torch.full((N, Tk, 3 * D), float("nan"), dtype=DT[dtype], device="cuda")
    ok(L.dic_attn_bwd(dtype, p(qd), p(dev(km)), p(dd), p(dq), N, Tk, H, 64, 0.0, 0, stream()), L)
    torch.cuda.synchronize()
    for name, sl in (("dQ", slice(0, D)), ("dK", slice(D, 2 * D)), ("dV", slice(2 * D, 3 * D))):
        e = relerr(dq.float()[..., sl], qq.grad[..., sl])
        assert e < (2e-5 if dtype == F32 else 3e-2), f"{name} relerr {e}"


@pytest.mark.parametrize("dtype,Tk", [(F32, 18), (BF16, 18), (BF16, 34), (BF16, 64)])
def test_attention_dropout_mask_is_shared_by_forward_and_backward(L, dtype, Tk):
    N, H, D, pd, seed = 2, 12, 768, 0.3, 777
    g = torch.Generator().manual_seed(4)
    qkv = torch.randn(N, Tk, 3 * D, generator=g)
    # V = identity block per head -> ctx[i][d=j] = P_dropped[i][j]
    v = torch.zeros(N, Tk, H, 64)
    for j in range(Tk):
        v[:, j, :, j] = 1.0
    qkv[..., 2 * D:] = v.reshape(N, Tk, D)
    km = torch.ones(N, Tk, dtype=torch.uint8)
    qd = dev(qkv, DT[dtype])
    ctx = torch.zeros(N, Tk, D, dtype=DT[dtype], device="cuda")
    ok(L.dic_attn_fwd(dtype, p(qd), p(dev(km)), p(ctx), N, Tk, H, 64, pd, seed, stream()), L)
    Pd = ctx.float().cpu().reshape(N, Tk, H, 64)[..., :Tk]          # [n, i, h, j]
    drop_rate = float((Pd == 0).float().mean())
    assert abs(drop_rate - pd) < 0.04
    # backward with dO = 1: dV[j][d] = sum_i P_dropped[i][j]
    dq = torch.zeros(N, Tk, 3 * D, dtype=DT[dtype], device="cuda")
    ones = torch.ones(N, Tk, D, dtype=DT[dtype], device="cuda")
    ok(L.dic_attn_bwd(dtype, p(qd), p(dev(km)), p(ones), p(dq), N, Tk, H, 64, pd, seed, stream()), L)
    dV = dq.float().cpu()[..., 2 * D:].reshape(N, Tk, H, 64)[..., 0]  # [n, j, h]
    colsum = Pd.sum(1).permute(0, 2, 1)                               # [n, j, h]
    assert relerr(dV, colsum) < (1e-5 if dtype == F32 else 2e-2)


# ------------------------------------------------------------------------------------------------ losses & small kernels
@pytest.mark.parametrize("kind,name", list(enumerate(["series_sum_sample_mean", "series_sum", "mse_series_mean", "mse_series_sum"])))
def test_emb_loss_kinds(L, kind, name):
    N, B, Lq, Tk = 6, 3, 16, 18
    g = torch.Generator().manual_seed(kind)
    xo = torch.randn(N, Tk, 768, generator=g)
    x0 = torch.randn(B, Lq, 768, generator=g)
    cfg = R.Config(BATCH_SIZE=B)
    xx = xo.double().requires_grad_(True)
    ref = R.LOSS_FUNCS[name](xx[:, :Lq], x0.double().repeat(2, 1, 1), cfg)
    ref.backward()
    scale = {0: 1 / (N * 768), 1: 1 / B / 768 / 100, 2: 1 / N, 3: 1 / B}[kind]
    per = torch.zeros(N, device="cuda")
    dx = torch.full((N, Tk, 768), float("nan"), device="cuda")
    gs = torch.full((N,), scale, device="cuda")
    xr = torch.zeros(N * Lq, 768, device="cuda")
    out = torch.zeros(4, device="cuda")
    ok(L.dic_emb_loss(F32, kind, p(dev(xo)), p(dev(x0)), B, p(per), p(dx), p(gs), p(xr), N, Lq, Tk, 768, stream()), L)
    ok(L.dic_seg_sum(p(per), N, N, scale, 0.0, p(out), 0, stream()), L)
    torch.cuda.synchronize()
    assert abs(float(out[0]) - float(ref)) < 2e-6 * abs(float(ref))
    assert relerr(dx, xx.grad) < 1e-5
    assert torch.equal(xr.cpu().reshape(N, Lq, 768), xo[:, :Lq])


def test_small_kernels(L):
    g = torch.Generator().manual_seed(8)
    # add_rows
    N, Lq, Tk = 4, 16, 18
    dx = torch.randn(N, Tk, 768, generator=g)
    dxr = torch.randn(N * Lq, 768, generator=g)
    d = dev(dx)
    ok(L.dic_add_rows(p(d), p(dev(dxr)), N, Lq, Tk, 768, stream()), L)
    ref = dx.clone()
    ref[:, :Lq] += dxr.reshape(N, Lq, 768)
    assert torch.allclose(d.cpu(), ref)
    # cfg mix fwd/bwd
    x = torch.randn(7, Tk * 768, generator=g)
    gi = torch.tensor([1, 4])
    xd = dev(x)
    ok(L.dic_cfg_mix_fwd(p(xd), p(xd[5:]), p(dev(gi)), 2, Tk * 768, 0.3, stream()), L)
    ref = x.clone()
    ref[gi] = 1.3 * x[5:7] - 0.3 * x[gi]
    assert torch.allclose(xd.cpu(), ref, atol=1e-6)
    dxx = torch.randn(7, Tk * 768, generator=g)
    dd = dev(dxx)
    ok(L.dic_cfg_mix_bwd(p(dd), p(dd[5:]), p(dev(gi)), 2, Tk * 768, 0.3, stream()), L)
    ref = dxx.clone()
    ref[5:7] = 1.3 * dxx[gi]
    ref[gi] = -0.3 * dxx[gi]
    assert torch.allclose(dd.cpu(), ref, atol=1e-6)
    # colsum (bf16 and f32 inputs, ld > cols, accumulate)
    a = torch.randn(1000, 3072, generator=g)
    out = torch.ones(2304, device="cuda")
    ws = torch.zeros(64 * 2304, device="cuda")
    ok(L.dic_colsum(F32, p(dev(a)), 1000, 2304, 3072, p(out), 1, p(ws), stream()), L)
    assert relerr(out, 1 + a[:, :2304].double().sum(0)) < 1e-5
    ab = dev(a, torch.bfloat16)
    out = torch.zeros(3072, device="cuda")
    ok(L.dic_colsum(BF16, p(ab), 1000, 3072, 3072, p(out), 0, p(dev(torch.zeros(64 * 3072))), stream()), L)
    assert relerr(out, ab.float().cpu().double().sum(0)) < 1e-5
    # seq_sum
    y = torch.randn(5, 16, 768, generator=g)
    fl = torch.tensor([1, 0, 0, 1, 1], dtype=torch.uint8)
    oa, of = torch.zeros(5, 768, device="cuda"), torch.zeros(5, 768, device="cuda")
    ok(L.dic_seq_sum(p(dev(y)), p(dev(fl)), p(oa), p(of), 5, 16, 768, stream()), L)
    assert torch.allclose(oa.cpu(), y.sum(1), atol=1e-5) and torch.allclose(of.cpu(), y.sum(1) * fl[:, None].float(), atol=1e-5)


def test_adamw_matches_oracle_and_writes_bf16_shadow(L):
    n = 4096 + 64
    g = torch.Generator().manual_seed(21)
    p0 = torch.randn(n, generator=g)
    ref_p = p0.clone().requires_grad_(True)
    opt = R.AdamW([ref_p], lr=1e-4)
    P, M_, V_ = dev(p0), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    sh = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
    for step in range(1, 4):
        grad = torch.randn(n, generator=g) * (10.0 ** (step - 3))
        ref_p.grad = grad.clone()
        opt.step()
        ok(L.dic_adamw(p(P), p(dev(grad * 4)), p(M_), p(V_), p(sh), n, 1e-4, 0.9, 0.999, 1e-8, 0.01, 1 - 0.9 ** step, 1 - 0.999 ** step, 0.25, stream()), L)
        torch.cuda.synchronize()
        np.testing.assert_allclose(P.cpu().numpy(), ref_p.detach().numpy(), rtol=0, atol=1e-6)
    assert torch.equal(sh.cpu(), P.cpu().to(torch.bfloat16))
    # split-weight form: the same update, plus lo = bf16(p - bf16(p))
    P2, M2, V2 = dev(p0), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    P1, M1, V1 = dev(p0), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    sh2, lo2 = torch.zeros(n, dtype=torch.bfloat16, device="cuda"), torch.zeros(n, dtype=torch.bfloat16, device="cuda")
    grad = dev(torch.randn(n, generator=g))
    ok(L.dic_adamw_hl(p(P2), p(grad), p(M2), p(V2), p(sh2), p(lo2), n, 1e-4, 0.9, 0.999, 1e-8, 0.01, 0.1, 0.001, 1.0, stream()), L)
    ok(L.dic_adamw(p(P1), p(grad), p(M1), p(V1), 0, n, 1e-4, 0.9, 0.999, 1e-8, 0.01, 0.1, 0.001, 1.0, stream()), L)
    torch.cuda.synchronize()
    assert torch.equal(P1, P2) and torch.equal(M1, M2) and torch.equal(V1, V2)
    assert torch.equal(sh2.cpu(), P2.cpu().to(torch.bfloat16)) and torch.equal(lo2.cpu(), (P2.cpu() - sh2.cpu().float()).to(torch.bfloat16))
    assert float((sh2.float() + lo2.float() - P2).abs().max() / P2.abs().max()) < 2e-5


def test_step_prep_randint_zero(L):
    """The step-input kernel against the repeat / hstack / cat statements of ref :406-415, 426, 434-437; the timestep draw; the range fill."""
    S, B, Lq = 3, 4, 16
    g = torch.Generator().manual_seed(2)
    img, txt = torch.randn(B, 512, generator=g), torch.randn(B, 512, generator=g)
    mask = (torch.rand(B, Lq, generator=g) > 0.3).long()
    ids = torch.randint(0, 30522, (B, Lq), generator=g)
    N = S * B + B
    for Tk in (Lq + 2, Lq + 1, Lq):
        img_in, txt_in = torch.full((N, 512), float("nan"), device="cuda"), torch.full((N, 512), float("nan"), device="cuda")
        km = torch.full((N, Tk), 9, dtype=torch.uint8, device="cuda")
        at = torch.full((N,), 9, dtype=torch.uint8, device="cuda")
        tgt = torch.full((N * Lq,), -5, dtype=torch.int64, device="cuda")
        gs = torch.full((N,), float("nan"), device="cuda")
        ok(L.dic_step_prep(p(dev(img)), p(dev(txt)), p(dev(mask)), p(dev(ids)), S, B, Lq, Tk, p(img_in), p(txt_in), p(km), p(at), p(tgt), p(gs),
                           0.25, 0.5, stream()), L)
        torch.cuda.synchronize()
        assert torch.equal(img_in.cpu(), torch.cat([img.repeat(S, 1), img])) and torch.equal(txt_in.cpu(), torch.cat([txt.repeat(S, 1), txt]))
        m = (mask != 0).to(torch.uint8)
        one = torch.ones(B, 1, dtype=torch.uint8)
        row = torch.cat([m, one, 0 * one][: 1 + (Tk - Lq)], 1)
        assert torch.equal(km.cpu(), torch.cat([row.repeat(S, 1), row]))
        assert bool((at == 0).all()) and torch.equal(tgt.cpu(), torch.cat([ids.repeat(S, 1), ids]).reshape(-1))
        assert torch.equal(gs.cpu(), torch.tensor([0.25] * (S * B) + [0.5] * B))
    t = torch.full((4096,), -1, dtype=torch.int64, device="cuda")
    ok(L.dic_randint(p(t), 4096, 100, 77, stream()), L)
    t2 = torch.empty_like(t)
    ok(L.dic_randint(p(t2), 4096, 100, 77, stream()), L)
    tc = t.cpu()
    assert torch.equal(tc, t2.cpu()) and int(tc.min()) == 0 and int(tc.max()) == 99 and abs(float(tc.float().mean()) - 49.5) < 2.0
    assert torch.bincount(tc, minlength=100).min() > 15
    z = torch.full((1000,), 3.0, device="cuda")
    ok(L.dic_zero(p(z) + 16, 16 * 50, stream()), L)
    torch.cuda.synchronize()
    assert bool((z[4:204] == 0).all()) and bool((z[:4] == 3).all()) and bool((z[204:] == 3).all())


@pytest.mark.parametrize("T,cu_cap", [(64 * 11 + 5, 0), (64 * 40, 0), (64 * 3 + 7, 0), (64 * 70 + 1, 7), (64 * 40, 10), (64 * 11 + 5, 5)])
def test_wgrad_group_matches_separate_products(L, T, cu_cap):
    """dic_wgrad_group: several dW = dY^T X (+ db = colsum dY) in one split-K launch + fold, against fp64; bit-identical when repeated; a small
    cu_cap makes every workgroup walk many (slice, tile) units.  The last two cases (as many CUs as tiles / half as many) take ONE K-slice per
    tile: written in place, no slab and no fold (round 5: the two-layer weight-gradient launch of the training step)."""
    shapes = [(768, 256, True), (256, 768, False), (512, 264, True)]            # (M, N, with bias gradient)
    g = torch.Generator().manual_seed(T)
    items, keep, refs = [], [], []
    for M, N, has_b in shapes:
        dY, X = torch.randn(T, M, generator=g), torch.randn(T, N, generator=g)
        dYd, Xd = dev(dY, torch.bfloat16), dev(X, torch.bfloat16)
        dW = torch.full((M, N), float("nan"), device="cuda")
        db = torch.full((M,), float("nan"), device="cuda") if has_b else None
        keep += [dYd, Xd, dW, db]
        items.append(dic._lib.WgradItem(dY=p(dYd), ldy=M, X=p(Xd), ldx=N, dW=p(dW), db=p(db), M=M, N=N))
        refs.append((dYd.float().cpu().double().t() @ Xd.float().cpu().double(), dYd.float().cpu().double().sum(0), dW, db))
    arr = (dic._lib.WgradItem * len(items))(*items)
    nbytes = L.dic_wgrad_group_ws_bytes(arr, len(items), T, cu_cap)
    assert nbytes > 0
    ws = torch.empty(nbytes // 4, device="cuda")
    outs = []
    for _ in range(2):
        ok(L.dic_wgrad_group(arr, len(items), T, p(ws), nbytes, cu_cap, stream()), L)
        torch.cuda.synchronize()
        outs.append([(r[2].clone(), None if r[3] is None else r[3].clone()) for r in refs])
        ws.fill_(float("nan"))                                                     # nothing stale may be folded
    for (w_ref, b_ref, dW, db), (w0, b0), (w1, b1) in zip(refs, outs[0], outs[1]):
        assert relerr(w0, w_ref) < 2e-6 and torch.equal(w0, w1)
        if db is not None:
            assert relerr(b0, b_ref) < 2e-6 and torch.equal(b0, b1)
    # too small a workspace is refused
    assert L.dic_wgrad_group(arr, len(items), T, p(ws), nbytes - 16, cu_cap, stream()) != 0

"""TRAIN_EMBEDDING ablation of the reference (CLIP-DDPM.py:98-102, 238-243, 260-262, 292-293, 319-320, 325-327).

Instead of DistilBERT's frozen 768-d token embedding the model learns a `IN_CHANNEL` = 16-d embedding and rounding head;
`input_projection` (16 -> 768) and `output_projection` (768 -> 16) sit around the unchanged encoder, diffusion and both
losses live in the 16-d space, and x_0 = embedding(ids) carries gradient (into q_sample and as the loss target).

Not a hot path of the metric: the 16-wide GEMMs run on the exact-fp32 MFMA kernel with K padded to its 32-deep step, whatever
the encoder's dtype; everything else reuses the kernels of the main path (q_sample, encoder forward/backward, embedding
losses, streaming CE / dlogits, AdamW over the same flat buffers).  Every switch of the main path applies: concat / add
fusion, the four loss functions, x_0 or x_{t-1} prediction, classifier-free guidance (guided copies stacked into the batch).
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import DIC_F32
from .config import cfg

LOSS_RING = 4096          # same lifetime of returned loss tensors as diffusion.LOSS_RING

KP = 32          # K-step of the fp32 MFMA GEMM: the 16-d operands are zero-padded to it


def _p(t):
    return t.data_ptr()


def _check_config(model):
    assert cfg.IN_CHANNEL == model.params.in_channel, "cfg.IN_CHANNEL changed after the model was built"


def _buffers(model, N, L, Tk, M):
    """Scratch of the 16-d side, cached per batch geometry."""
    key = (N, L, Tk, M)
    b = model._te_ws.get(key)
    if b is None:
        model._evict(model._te_ws, 2)
        dev, C = model.device, model.params.in_channel

        def f(*s):
            return torch.zeros(*s, dtype=torch.float32, device=dev)
        b = dict(x16=f(N, L, C), x16p=f(N * L, KP), Winp=f(768, KP), x_out16=f(N, Tk, C), Woutp=f(KP, 768), dx16=f(N, Tk, C), g16=f(N, Tk, C),
                 dx16p=f(N * Tk, KP), xr16=f(M, C), xr32=f(M, KP), Wlm32=f(model.vpad, KP), dxr32=f(M, KP), dxr16=f(M, C),
                 x16Tk=f(N, Tk, C), dxin16=f(N, Tk, C), dx0=f(cfg.BATCH_SIZE, L, C), dlogits=None, per_seq=f(N), gscale=f(N), ring=f(LOSS_RING, 8), slot=0,
                 cs=f(64 * max(Tk * C, 768)))
        model._te_ws[key] = b
    return b


def refresh_padded_weights(model, b):
    """Padded operand copies of the three 16-wide weights (the parameters themselves stay [.,16] in the flat buffer)."""
    P, C = model.params, model.params.in_channel
    b["Winp"][:, :C].copy_(P.slot_view(P.P, "Win"))
    b["Woutp"][:C].copy_(P.slot_view(P.P, "Wout"))
    b["Wlm32"][:, :C].copy_(P.slot_view(P.P, "Wlm16"))


def project_in(model, b, x16, N, L):
    """:292-293  x = input_projection(x): [N,L,16] -> the encoder's input buffer [N,L,768]."""
    C = model.params.in_channel
    ws = model._workspace(N, L, b["drop_txt"])
    b["x16"].copy_(x16)
    b["x16p"][:, :C].copy_(x16.reshape(N * L, C))
    o, P = model.ops, model.params
    o.begin()
    o.gemm(_p(b["x16p"]), _p(b["Winp"]), _p(ws["xin"]), N * L, 768, KP, KP, KP, 768, bias=P.ptr("bin"), out_f32=1, dtype=DIC_F32)
    return ws["xin"]


def project_out(model, b, x_out768, N, Tk):
    """:319-320  x_out = output_projection(x_out): [N,Tk,768] -> [N,Tk,16]."""
    C = model.params.in_channel
    o, P = model.ops, model.params
    o.begin()
    o.gemm(_p(x_out768), P.ptr("Wout"), _p(b["x_out16"]), N * Tk, C, 768, 768, 768, C, bias=P.ptr("bout"), out_f32=1, dtype=DIC_F32)
    return b["x_out16"]


def _masks(model, mask, S, B, L, drop_txt, gi):
    """Key masks of the stacked batch [x_t rows | guided copies | x_1 rows] (ref :296-297, 309-311)."""
    dev = model.device
    m = (mask.to(dev) != 0).to(torch.uint8)
    m_rep = m.repeat(S, 1)
    if model.concat:
        one_t = torch.ones(S * B, 1, dtype=torch.uint8, device=dev)
        one_b = torch.ones(B, 1, dtype=torch.uint8, device=dev)
        pt = torch.cat([m_rep, one_t] if drop_txt else [m_rep, one_t, 0 * one_t], 1)
        pb = torch.cat([m, one_b] if drop_txt else [m, one_b, 0 * one_b], 1)
        if gi is not None:
            return torch.cat([pt, torch.cat([m_rep[gi], one_t[:len(gi)], one_t[:len(gi)]], 1), pb])
        return torch.cat([pt, pb])
    return torch.cat([m_rep, m_rep[gi], m]) if gi is not None else torch.cat([m_rep, m])


def loss(model, x_t, x_1, x_tgt, x_0, image_clip, text_clip, mask, idx, kind, cfg_uniform=None):
    from .diffusion import _guidance_uniform
    """`loss` (ref :382-445) with the learned embedding: returns the three loss scalars and, under grad mode, leaves the encoder's
    output gradient in the workspace (so `model.backward()` runs as usual) plus what `backward_tail` needs."""
    _check_config(model)
    S, B, L, C = cfg.SAMPLE_SIZE, cfg.BATCH_SIZE, cfg.MAX_LENGTH, cfg.IN_CHANNEL
    dev, lib = model.device, _lib.lib()
    Nt = S * B
    w = float(cfg.CLASSIFIER_FREE_WEIGHT)
    want_grad = torch.is_grad_enabled()
    # ---- classifier-free-guidance draw (ref :406-412), as in diffusion.loss
    gi = None
    if w > 0:
        u = cfg_uniform.to(dev) if cfg_uniform is not None else _guidance_uniform(Nt, dev)
        cm = (u > cfg.CLASSIFIER_FREE_PROB).reshape(Nt)
        if model.rank_rows_forced:
            cm[0] = False
            cm[1] = True
        gi = cm.nonzero().squeeze(1)
        if gi.numel() == 0:
            gi = None
    Ng = 0 if gi is None else int(gi.numel())
    N = Nt + Ng + B
    M = (Nt + B) * L
    drop_txt = model.concat and Ng == 0 and cfg.DROP_UNUSED_TEXT_ROW
    ws = model._workspace(N, L, drop_txt)
    Tk = ws["Tk"]
    b = _buffers(model, N, L, Tk, M)
    b["drop_txt"] = drop_txt
    model.params.text_unused = (not model.concat) and Ng == 0
    refresh_padded_weights(model, b)
    x_t = x_t.to(dev, torch.float32)
    x16 = torch.cat([x_t, x_t[gi], x_1.to(dev, torch.float32)]) if Ng else torch.cat([x_t, x_1.to(dev, torch.float32)])
    xin = project_in(model, b, x16, N, L)
    img = image_clip.to(dev, torch.float32)
    txt = text_clip.to(dev, torch.float32)
    img_rep, txt_rep = img.repeat(S, 1), txt.repeat(S, 1)
    ic = torch.cat([img_rep, img_rep[gi], img]) if Ng else torch.cat([img_rep, img])
    tc = torch.cat([txt_rep, txt_rep[gi], txt]) if Ng else torch.cat([txt_rep, txt])
    km = _masks(model, mask, S, B, L, drop_txt, gi)
    add_txt = torch.zeros(N, dtype=torch.uint8, device=dev)
    if Ng:
        add_txt[Nt:Nt + Ng] = 1
    x_out768 = model.encode(xin, ic, tc, km, add_txt, drop_txt=drop_txt)
    st = model.ops.stream
    if Ng:          # ref :314-317: the mix happens on the encoder output, before output_projection
        _lib.check(lib.dic_cfg_mix_fwd(_p(x_out768), _p(x_out768) + Nt * Tk * 768 * 4, _p(gi), Ng, Tk * 768, w, st), "cfg_mix_fwd")
    x_out16 = project_out(model, b, x_out768, N, Tk)

    # ---- embedding losses on the 16-d output (ref :77-87: the mean / the literal 768 of `series_sum` are the reference's)
    if kind == 0:
        sa, sb = 1.0 / (Nt * C), 1.0 / (B * C)
    elif kind == 1:
        sa = sb = 1.0 / cfg.BATCH_SIZE / 768 / 100
    elif kind == 2:
        sa, sb = 1.0 / Nt, 1.0 / B
    else:
        sa = sb = 1.0 / cfg.BATCH_SIZE
    if not cfg.USE_X_T_LOSS:
        sa = 0.0
    if not cfg.USE_X_1_LOSS:
        sb = 0.0
    b["gscale"][:Nt].fill_(sa)
    b["gscale"][Nt:].fill_(sb)
    x_0c = x_0.to(dev, torch.float32).contiguous()
    if cfg.X_0_PREDICTION:
        tgt_t, tgt_rows = x_0c, B
    else:
        assert x_tgt.shape == x_t.shape
        tgt_t, tgt_rows = x_tgt.to(dev, torch.float32).contiguous(), Nt
    dx = _p(b["dx16"]) if want_grad else 0
    row = Tk * C
    if want_grad and Ng:
        b["dx16"][Nt:Nt + Ng].zero_()              # the guided copies' own outputs feed nothing after the mix
    _lib.check(lib.dic_emb_loss(DIC_F32, kind, _p(x_out16), _p(tgt_t), tgt_rows, _p(b["per_seq"]), dx, _p(b["gscale"]), _p(b["xr16"]), Nt, L, Tk,
                                C, st), "emb_loss")
    off = (Nt + Ng) * row * 4
    _lib.check(lib.dic_emb_loss(DIC_F32, kind, _p(x_out16) + off, _p(x_0c), B, _p(b["per_seq"]) + Nt * 4, (dx + off) if want_grad else 0,
                                _p(b["gscale"]) + Nt * 4, _p(b["xr16"]) + Nt * L * C * 4, B, L, Tk, C, st), "emb_loss")
    b["slot"] = (b["slot"] + 1) % LOSS_RING     # ring of result slots, as diffusion.loss (same lifetime: LOSS_RING calls)
    out = b["ring"][b["slot"]]
    _lib.check(lib.dic_seg_sum(_p(b["per_seq"]), Nt + B, Nt, sa, sb, _p(out), 0, st), "seg_sum")
    if want_grad:
        b["g16"].copy_(b["dx16"])                  # the loss gradient alone: its negative flows into the targets

    # ---- rounding loss on the learned head (ref :323, 432-445)
    if cfg.USE_PROB_LOSS:
        cw = model._ce_workspace(M)
        ids = idx.to(dev, torch.int64)
        cw["tgt"][:Nt * L].copy_(ids.repeat(S, 1).reshape(-1))
        cw["tgt"][Nt * L:].copy_(ids.reshape(-1))
        b["xr32"][:, :C].copy_(b["xr16"])
        o = model.ops
        o.gemm(_p(b["xr32"]), _p(b["Wlm32"]), 0, M, model.vocab, KP, KP, KP, 0, epi=_EPI_CE_PARTIAL, tgt=_p(cw["tgt"]),
               partial=_p(cw["partial"]), tgt_logit=_p(cw["tgt_logit"]), dtype=DIC_F32, tile=128)
        _lib.check(lib.dic_ce_combine(_p(cw["partial"]), _p(cw["tgt_logit"]), M, lib.dic_ce_n_partials(model.vocab, 128), _p(cw["lse"]),
                                      _p(cw["argmax"]), _p(cw["nll"]), st), "ce_combine")
        ca = (1.0 / Nt) if kind in (0, 2) else (1.0 / cfg.BATCH_SIZE)
        cb = (1.0 / B) if kind in (0, 2) else (1.0 / cfg.BATCH_SIZE)
        rw = float(cfg.ROUNDING_WEIGHT)
        _lib.check(lib.dic_seg_sum(_p(cw["nll"]), M, Nt * L, rw * ca, rw * cb, _p(out) + 4 * 4, 0, st), "seg_sum")
        if want_grad:
            if b["dlogits"] is None:
                b["dlogits"] = torch.empty(M, model.vpad, dtype=torch.float32, device=dev)
            o.gemm(_p(b["xr32"]), _p(b["Wlm32"]), _p(b["dlogits"]), M, model.vocab, KP, KP, KP, model.vpad, epi=_EPI_CE_DLOGITS, tgt=_p(cw["tgt"]),
                   lse=_p(cw["lse"]), ce_rows_a=Nt * L, ce_scale_a=rw * ca, ce_scale_b=rw * cb, dtype=DIC_F32, tile=128)
            # d(lm_head.weight) = dlogits^T x_out16 ; d x_out16 += dlogits W_lm
            P = model.params
            o.gemm(_p(b["dlogits"]), _p(b["xr16"]), P.ptr("Wlm16", "G"), model.vpad, C, M, model.vpad, C, C, a_km=1, b_km=1, out_f32=1, dtype=DIC_F32)
            o.gemm(_p(b["dlogits"]), _p(b["Wlm32"]), _p(b["dxr32"]), M, KP, model.vpad, model.vpad, KP, KP, b_km=1, out_f32=1, dtype=DIC_F32)
            b["dxr16"].copy_(b["dxr32"][:, :C])
            _lib.check(lib.dic_add_rows(_p(b["dx16"]), _p(b["dxr16"]), Nt, L, Tk, C, st), "add_rows")
            _lib.check(lib.dic_add_rows(_p(b["dx16"]) + off, _p(b["dxr16"]) + Nt * L * C * 4, B, L, Tk, C, st), "add_rows")
        prob = out[6]
    else:
        prob = torch.zeros((), dtype=torch.float32, device=dev)
        if want_grad:
            model.params.slot_view(model.params.G, "Wlm16").zero_()

    if want_grad:
        # ---- output_projection backward: dW, db, and the encoder's output gradient dx768 = dx16 W_out
        o, P = model.ops, model.params
        T = N * Tk
        o.gemm(_p(b["dx16"]), _p(x_out768), P.ptr("Wout", "G"), C, 768, T, C, 768, 768, a_km=1, b_km=1, out_f32=1, dtype=DIC_F32)
        _lib.check(lib.dic_colsum(DIC_F32, _p(b["dx16"]), T, C, C, P.ptr("bout", "G"), 0, _p(b["cs"]), st), "colsum")
        b["dx16p"][:, :C].copy_(b["dx16"].reshape(T, C))
        o.gemm(_p(b["dx16p"]), _p(b["Woutp"]), _p(ws["dx_out"]), T, 768, KP, KP, 768, 768, b_km=1, out_f32=1, dtype=DIC_F32)
        if Ng:
            _lib.check(lib.dic_cfg_mix_bwd(_p(ws["dx_out"]), _p(ws["dx_out"]) + Nt * Tk * 768 * 4, _p(gi), Ng, Tk * 768, w, st), "cfg_mix_bwd")
        model._te_pending = dict(b=b, N=N, L=L, Tk=Tk, S=S, B=B, ids=idx.to(dev, torch.int64), gi=gi, Ng=Ng)
    model._pending = want_grad
    return out[0], out[1], prob


def backward_tail(model, t, t_next=None):
    """After `model.backward()`: input_projection, q_sample and embedding backward (ref :459-468 under autograd).
    t_next: the second timestep vector of x_{t-1} prediction (the x_t loss then targets diffuse_t(x_0, t_next), ref :364-380)."""
    st8 = model._te_pending
    model._te_pending = None
    b, N, L, Tk, S, B = st8["b"], st8["N"], st8["L"], st8["Tk"], st8["S"], st8["B"]
    C = model.params.in_channel
    o, P, lib = model.ops, model.params, _lib.lib()
    o.begin()
    st = o.stream
    ws = model._saved
    dy0 = ws["dy0"]                                     # [N,Tk,768]: gradient wrt the fused rows; rows t < L = d(input_projection(x))
    T = N * Tk
    b["x16Tk"].zero_()
    b["x16Tk"][:, :L].copy_(b["x16"])
    o.gemm(_p(dy0), _p(b["x16Tk"]), P.ptr("Win", "G"), 768, C, T, 768, C, C, a_km=1, b_km=1, out_f32=1, dtype=DIC_F32)
    gpos = P.ptr("pos", "G")                            # dpos[t] = sum_n dy0[n][t]: its rows t < L add up to d(input_projection.bias)
    _lib.check(lib.dic_colsum(DIC_F32, gpos, L, 768, 768, P.ptr("bin", "G"), 0, _p(b["cs"]), st), "colsum")
    o.gemm(_p(dy0), P.ptr("Win"), _p(b["dxin16"]), T, C, 768, 768, C, C, b_km=1, out_f32=1, dtype=DIC_F32)
    from . import diffusion
    diffusion.alpha_cumprod_table(model.device)
    gi, Ng = st8["gi"], st8["Ng"]
    if Ng:          # a guided copy is the same x_t row projected a second time: its input gradient belongs to that row
        b["dxin16"].index_add_(0, gi, b["dxin16"][S * B:S * B + Ng].clone())
    tt = t.to(model.device, torch.int64).reshape(-1).contiguous()
    tn = t_next.to(model.device, torch.int64).reshape(-1).contiguous() if t_next is not None else None
    _lib.check(lib.dic_te_dx0(_p(b["dxin16"]), _p(b["g16"]), _p(diffusion._state["sqrt_ac"]), _p(tt), _p(tn) if tn is not None else 0, S, B, L, Tk,
                              C, cfg.STEP_TOT, S * B + Ng, _p(b["dx0"]), st), "te_dx0")
    ids = st8["ids"].reshape(-1)
    sorted_ids, order = torch.sort(ids, stable=True)
    gE = P.slot_view(P.G, "E16")
    gE.zero_()
    _lib.check(lib.dic_embed_scatter(_p(sorted_ids), _p(order), _p(b["dx0"]), ids.numel(), C, model.vocab, _p(gE), st), "embed_scatter")


def forward(model, x, image_clip, text_clip, mask, concat_mask, with_logits=True):
    """`model(...)` (ref :271-323) with the projections around the encoder: returns (logits [n,L,V], x_out [n,Tk,16])."""
    _check_config(model)
    n, L, C = x.shape[0], cfg.MAX_LENGTH, cfg.IN_CHANNEL
    dev = model.device
    Tk = L + 2 if model.concat else L
    b = _buffers(model, n, L, Tk, n * L)
    b["drop_txt"] = False
    refresh_padded_weights(model, b)
    xin = project_in(model, b, x.to(dev, torch.float32), n, L)
    m = (mask.to(dev) != 0).to(torch.uint8)
    if model.concat:
        one = torch.ones(n, 1, dtype=torch.uint8, device=dev)
        km = torch.cat([m, one, 0 * one], 1)
    else:
        km = m
    x768 = model.encode(xin, image_clip.to(dev, torch.float32).reshape(n, 512), text_clip.to(dev, torch.float32).reshape(n, 512), km,
                        torch.zeros(n, dtype=torch.uint8, device=dev))
    x_out = project_out(model, b, x768, n, Tk).clone()
    logits = lm_head(model, x_out[:, :L, :]) if with_logits else None
    return logits, x_out


def lm_head(model, h):
    """ref :323 with the learned head: [.., 16] -> [.., V] fp32 (materialised: API use only)."""
    C = model.params.in_channel
    shp = h.shape[:-1]
    x = torch.zeros(h.numel() // C, KP, dtype=torch.float32, device=model.device)
    x[:, :C].copy_(h.reshape(-1, C))
    W = torch.zeros(model.vpad, KP, dtype=torch.float32, device=model.device)
    W[:, :C].copy_(model.params.slot_view(model.params.P, "Wlm16"))
    Vp = model.vocab + (-model.vocab) % 4
    out = torch.empty(x.shape[0], Vp, dtype=torch.float32, device=model.device)
    model.ops.begin()
    model.ops.gemm(_p(x), _p(W), _p(out), x.shape[0], Vp, KP, KP, KP, Vp, out_f32=1, dtype=DIC_F32)
    return out[:, :model.vocab].reshape(*shp, model.vocab)


@torch.no_grad()
def sample(model, image_clip, steps, start, return_hidden):
    """The sampling loop (ref :611-621) in the 16-d space."""
    _check_config(model)
    dev = model.device
    B, L, C = image_clip.shape[0], cfg.MAX_LENGTH, cfg.IN_CHANNEL
    restored = start.to(dev, torch.float32) if start is not None else torch.randn(B, L + 2, C, device=dev)
    img = image_clip.to(dev, torch.float32)
    x = restored[:, :L, :].contiguous()
    ones = torch.ones(B, L, device=dev)
    cm = torch.tensor([1, 0], device=dev).repeat(B, 1)
    zeros = torch.zeros_like(img).unsqueeze(1)
    x_out = None
    for _ in range(steps):
        _, x_out = forward(model, x, img.unsqueeze(1), zeros, ones, cm, with_logits=False)
        x = x_out[:, :L, :].contiguous()
    ids = lm_head(model, x).argmax(-1)
    return (ids, x_out) if return_hidden else ids


from .engine import EPI_CE_PARTIAL as _EPI_CE_PARTIAL, EPI_CE_DLOGITS as _EPI_CE_DLOGITS   # noqa: E402

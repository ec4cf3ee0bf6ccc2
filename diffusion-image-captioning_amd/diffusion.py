"""Diffusion schedule, q_sample, loss, training step, validation and sampling with the reference's call
signatures (ref CLIP-DDPM.py:337-501, 611-621), executed by the HIP kernels behind `engine.Denoiser`.

Signatures kept:
    diffuse_t(x, t) -> [t.numel()*B, L, C]            generate_diffuse_pair(x_0, t, t_next=None)
    loss(model, x_t, x_1, x_tgt, x_0, image_clip, text_clip, mask, idx, loss_func) -> (x_t_loss, x_1_loss, prob_loss)
    train_func(model, trainer, x, train=True) -> (l, x_t_loss, x_1_loss, prob_loss)
    validate(model) -> (val_x_t, val_x_1, val_prob)   sample(model, image_clip, steps=5) -> LongTensor[B, L]
Optional keyword-only arguments (`noise=`, `t=`, `noises=`, `cfg_uniform=`) inject the random draws so parity
tests can feed the exact values the CPU reference used; without them the draws come from the device RNG
(Philox inside the q_sample / dropout kernels, torch.randint/rand for the handful of host-side scalars).
"""
from __future__ import annotations

import math

import torch

from . import _lib
from .config import LOSS_KINDS, cfg
from .engine import Denoiser, _p

NOISE_SEED_BASE = 0xD1FF0000          # q_sample's Philox stream (parallel.configure_model_for_rank mixes the rank in)
GUIDANCE_SEED_BASE = 0xC1A55F4EE        # classifier-free-guidance uniforms (ref :407), drawn from a generator of their own
LOSS_RING = 4096                       # loss() returns views into a ring of result slots: valid until LOSS_RING later calls

_state = {"ac_key": None, "ac": None, "noise_seed": NOISE_SEED_BASE, "val_loader": None, "trainer": None, "cfg_gen": {}}


# ------------------------------------------------------------------ schedule (ref :337-346)
def alpha_cumprod_table(device=None) -> torch.Tensor:
    """alpha-bar[t], computed on the host exactly as the reference does (cosine: :338-342, linear: :344-346)."""
    ov = _state.get("ac_override")
    key = (cfg.COSIN_SCHEDULE, cfg.STEP_TOT, cfg.BETA_MIN, cfg.BETA_MAX, str(device), id(ov))
    if _state["ac_key"] != key:
        if ov is not None:
            assert ov.numel() == cfg.STEP_TOT, "alpha_cumprod override must have STEP_TOT entries"
            ac = ov
        elif cfg.COSIN_SCHEDULE:
            s = 0.008

            def sched(t):
                return torch.cos(math.pi / 2 * (t / cfg.STEP_TOT + s) / (1 + s)) ** 2
            ac = sched(torch.arange(cfg.STEP_TOT)) / sched(torch.zeros(1))
        else:
            betas = torch.hstack([torch.zeros(1), torch.linspace(cfg.BETA_MIN, cfg.BETA_MAX, cfg.STEP_TOT)])
            ac = torch.cumprod((1 - betas)[:-1], 0)
        dev = device if device is not None else "cuda:0"
        ac = ac.to(torch.float32)
        _state["ac"] = ac.to(dev).contiguous()
        # the two coefficient tables of q_sample (ref :360-361): CORRECTLY ROUNDED fp32 square roots (sqrt in fp64, then
        # round: exact for p' >= 2p+2).  torch's long-vector fp32 sqrt is not correctly rounded on every CPU, while the
        # reference takes sqrt of the S gathered entries (scalar path, correctly rounded) -- this matches the latter.
        _state["sqrt_ac"] = torch.sqrt(ac.double()).float().to(dev).contiguous()
        _state["sqrt_1mac"] = torch.sqrt((1 - ac).double()).float().to(dev).contiguous()
        _state["ac_key"] = key
    return _state["ac"]


def set_alpha_cumprod(table):
    """Override the alpha-bar table (custom schedules; also how the golden tests inject the reference host's table --
    `torch.cos` is not correctly rounded, so the cosine table differs by an ulp between CPU models).  None = rebuild
    from cfg on next use."""
    if table is None:
        _state["ac_key"] = None
        _state["ac_override"] = None
        return
    _state["ac_override"] = torch.as_tensor(table, dtype=torch.float32).cpu().contiguous()
    _state["ac_key"] = None


def seed_noise(seed: int):
    _state["noise_seed"] = int(seed)


def seed_guidance(seed: int, device=None):
    """(Re)seed the generator the classifier-free-guidance uniforms come from (per process = per rank under data parallelism).  It lives on
    the HOST: the step needs the number of guided rows for its launch shapes, and drawing the Nt uniforms on the CPU (a few hundred numbers)
    is what lets a guided step run without a device->host synchronisation.  `device` is accepted for compatibility and ignored."""
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)
    _state["cfg_gen"]["cpu"] = g
    _state["cfg_seed_value"], _state["cfg_draws"] = int(seed) & 0x7FFFFFFFFFFFFFFF, 0       # (what a checkpoint needs to re-derive another rank's stream)
    return g


def _guidance_gen():
    g = _state["cfg_gen"].get("cpu")
    if g is None:
        g = seed_guidance(GUIDANCE_SEED_BASE)
    return g


def _guidance_uniform(n, dev):
    """torch.rand((n, 1)) of ref :407 from the guidance generator (host), returned on `dev`."""
    g = _guidance_gen()
    _state["cfg_draws"] = _state.get("cfg_draws", 0) + int(n)
    return torch.rand((n, 1), generator=g).to(dev)


def _guidance_draw(model, Nt, cfg_uniform=None):
    """ref :406-412: which of the Nt x_t rows get a guided copy.  Host-side draw (see seed_guidance); the ascending row list goes to the device
    through a small ring of pinned staging buffers (asynchronous copy, no synchronisation).  Returns {"gi": device int64 [>= Ng], "Ng", "Nt"}."""
    dev = model.device
    if cfg_uniform is not None:                        # injected draw (parity tests): a host copy of it (this is the only path that may sync)
        u = cfg_uniform.detach().reshape(-1).to("cpu", torch.float32)
        assert u.numel() == Nt
    else:
        g = _guidance_gen()
        _state["cfg_draws"] = _state.get("cfg_draws", 0) + int(Nt)
        u = torch.rand(Nt, generator=g)
    cm = u > cfg.CLASSIFIER_FREE_PROB
    if model.rank_rows_forced:
        cm[0] = False
        cm[1] = True
    gi = cm.nonzero().squeeze(1)
    Ng = int(gi.numel())
    ring = _state.setdefault(("gi_ring", str(dev)), {"slots": [], "i": 0})
    if len(ring["slots"]) < 8:
        ring["slots"].append({"host": torch.empty(max(Nt, 1), dtype=torch.int64).pin_memory(), "dev": torch.empty(max(Nt, 1), dtype=torch.int64, device=dev),
                              "ev": None})
        slot = ring["slots"][-1]
    else:
        ring["i"] = (ring["i"] + 1) % 8
        slot = ring["slots"][ring["i"]]
        if slot["ev"] is not None:
            slot["ev"].synchronize()                   # its last upload (8 steps ago) has long finished
        if slot["host"].numel() < Nt:
            slot["host"] = torch.empty(Nt, dtype=torch.int64).pin_memory()
            slot["dev"] = torch.empty(Nt, dtype=torch.int64, device=dev)
    if Ng:
        slot["host"][:Ng].copy_(gi)
        slot["dev"][:Ng].copy_(slot["host"][:Ng], non_blocking=True)
        slot["ev"] = torch.cuda.Event()
        slot["ev"].record()
    return {"gi": slot["dev"], "Ng": Ng, "Nt": Nt}


T_SEED_BASE = 0x7157E9        # timestep draws: the SAME stream on every rank (one t-vector per step for the whole global batch, ref :461)


def _default_t_seed() -> int:
    return (T_SEED_BASE + (torch.initial_seed() & 0x3FFFFFFFFFFF)) & 0x7FFFFFFFFFFFFFFF


def _next_t_seed() -> int:
    """Seed of the next timestep draw: a per-process counter.  Its start follows torch's global seed (`torch.manual_seed(s)` changes the
    t-vectors, as it does for the reference's `torch.randint`, ref :460-461) unless `seed_timesteps` / `seed_all` set it.  Under data parallelism
    every rank must continue rank 0's counter (ref :461 shares one t-vector over the whole batch): that is ONE broadcast at an explicit
    synchronisation point -- `parallel.configure_model_for_rank` or the start of `harness.fit` (`parallel.share_timestep_seed`) -- never a lazy
    collective here: ranks that reach their first draw at different times (a rank-0-only sanity step, a restored checkpoint on some ranks)
    would hang in it without a hint (round-4 advisor).  A multi-rank draw from a counter that was never shared therefore raises."""
    if not _state.get("t_seed_shared", False):
        from . import parallel
        if parallel.world_size() > 1:
            raise RuntimeError("data-parallel run: the timestep stream was never shared between the ranks -- call parallel.configure_model_for_rank(model) "
                               "(or parallel.share_timestep_seed()) on EVERY rank before the first train_func / validate; torch.initial_seed() differs per "
                               "process, so the ranks would noise their shards at different timesteps (ref :461 draws one t-vector for the whole batch)")
    if "t_seed" not in _state:
        _state["t_seed"] = _default_t_seed()
    _state["t_seed"] += 1
    return _state["t_seed"]


def _draw_t(S, dev):
    """torch.randint(0, STEP_TOT, (S,1,1)) of ref :460-461 from a Philox kernel keyed by the step counter above."""
    seed = _next_t_seed()
    t = torch.empty((S, 1, 1), dtype=torch.int64, device=dev)
    _lib.check(_lib.lib().dic_randint(_p(t), S, int(cfg.STEP_TOT), seed, torch.cuda.current_stream().cuda_stream), "randint")
    return t


def seed_timesteps(seed: int):
    """Start the timestep stream at `seed`.  Data parallel: call it with the SAME value on every rank (that is what makes the stream shared;
    `parallel.assert_shared_timestep_seed` verifies it once per epoch) or let `parallel.share_timestep_seed` copy rank 0's."""
    _state["t_seed"] = int(seed) & 0x7FFFFFFFFFFFFFFF
    _state["t_seed_shared"] = True


def seed_all(seed: int, device=None):
    """One integer for every random stream of the step (timesteps, q_sample noise, guidance mask): what `torch.manual_seed(seed)` is to the
    reference's script.  Call it with the same value on every data-parallel rank BEFORE `parallel.configure_model_for_rank` (which then
    mixes the rank into the per-item streams and leaves the timestep stream shared).  Dropout masks follow the model's own seed."""
    s = int(seed) & 0xFFFFFFFFFFFF
    seed_timesteps(T_SEED_BASE + s * 0x9E3779B1)
    _state["noise_base"], _state["guidance_base"] = NOISE_SEED_BASE + s * 0x85EBCA6B, GUIDANCE_SEED_BASE + s * 0xC2B2AE35     # un-mixed: the
    seed_noise(_state["noise_base"])                                                                    # per-rank seeds derive from these
    seed_guidance(_state["guidance_base"], device)


def rng_state(model=None) -> dict:
    """Everything a resumed run needs to continue the SAME random streams (harness.save_checkpoint stores it): the timestep counter, the
    q_sample noise counter, the guidance generator and, given the model, its dropout-mask counter -- together with the data-parallel RANK
    they belong to: noise, dropout masks and guidance draws are per-rank streams (parallel.configure_model_for_rank), so a checkpoint written
    by one rank must not hand ITS streams to every rank that loads it (set_rng_state re-derives the others from it)."""
    from . import parallel
    _guidance_gen()
    st = {"t_seed": _state.get("t_seed"), "noise_seed": _state["noise_seed"], "rank": parallel.rank(),
          "guidance": {k: g.get_state().cpu() for k, g in _state["cfg_gen"].items()},
          "guidance_seed": _state.get("cfg_seed_value"), "guidance_draws": _state.get("cfg_draws", 0)}
    if model is not None:
        st["dropout_seed"] = int(model._seed)
    return st


def set_rng_state(st: dict, model=None):
    """Continue the streams of `rng_state`.  The timestep stream is shared by all ranks and restored as saved.  The per-item streams (noise,
    dropout, guidance) are restored exactly on the rank that saved them and SHIFTED by the rank difference on every other rank, the same way
    parallel.rank_seed separates them at start-up -- every rank loading rank 0's checkpoint keeps drawing its own eps / masks / guidance rows.
    (A guidance generator cannot be shifted: another rank reseeds it from the saved seed, the rank mix and the number of draws so far.)
    Checkpoints from before round 4 carry no rank (= 0) and may name the guidance generator 'cuda:0': it lives on the host now."""
    from . import parallel
    r, sr = parallel.rank(), int(st.get("rank", 0) or 0)
    shift = (r - sr) * parallel._RANK_MIX
    if st.get("t_seed") is not None:
        _state["t_seed"] = int(st["t_seed"])
        _state["t_seed_shared"] = True             # (every rank restores the same checkpoint's counter)
    _state["noise_seed"] = (int(st["noise_seed"]) + shift) & 0x7FFFFFFFFFFFFFFF       # mod 2^63, exactly like parallel.rank_seed at start-up
    gstates = list(st.get("guidance", {}).values())
    if r == sr and gstates:
        g = _guidance_gen()
        g.set_state(gstates[0])                        # (one generator, whatever device key an old checkpoint filed it under)
        _state["cfg_seed_value"], _state["cfg_draws"] = st.get("guidance_seed"), int(st.get("guidance_draws", 0) or 0)
    elif r != sr:
        base = st.get("guidance_seed")
        base = GUIDANCE_SEED_BASE if base is None else int(base)
        seed_guidance((base + int(st.get("guidance_draws", 0) or 0) * 0x9E3779B1 + shift) & 0x7FFFFFFFFFFFFFFF)
    if model is not None and "dropout_seed" in st:
        model._seed = (int(st["dropout_seed"]) + shift) & 0x7FFFFFFFFFFFFFFF


def _t_one(dev):
    """The constant t = 1 of the x_1 pass (ref :468), kept on the device."""
    key = ("t_one", str(dev))
    if key not in _state:
        _state[key] = torch.ones(1, dtype=torch.int64, device=dev)
    return _state[key]


def _next_seed():
    # the counter lives mod 2^63 like every per-rank seed (parallel.rank_seed): a stream restored from another rank's checkpoint
    # (set_rng_state) then continues bit for bit where an uninterrupted run of this rank would be
    _state["noise_seed"] = (_state["noise_seed"] + 0x9E3779B1) & 0x7FFFFFFFFFFFFFFF
    return _state["noise_seed"]


# ------------------------------------------------------------------ q_sample (ref :347-362)
def diffuse_t(x, t, *, noise=None, out=None):
    """x [B,L,C] fp32, t [S,...] int64 -> [S*B, L, C]; ONE noise tensor per call shared by all S (ref :359)."""
    _lib.require_gpu()
    L = _lib.lib()
    b, seq_len, c = x.shape
    dev = x.device
    x = x.contiguous()
    t = t.to(dev, torch.int64).reshape(-1).contiguous()
    S = t.numel()
    alpha_cumprod_table(dev)
    if out is None:
        out = torch.empty(S * b, seq_len, c, dtype=torch.float32, device=dev)
    nz = noise.to(dev, torch.float32).contiguous() if noise is not None else None
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.dic_qsample(_p(x), _p(nz), _p(t), _p(_state["sqrt_ac"]), _p(_state["sqrt_1mac"]), _p(out), 0, S, b, seq_len * c,
                             cfg.STEP_TOT, _next_seed(), st), "qsample")
    return out


def generate_diffuse_pair(x_0, t, t_next=None, *, noises=(None, None)):
    """ref :364-380."""
    if cfg.X_0_PREDICTION:
        return diffuse_t(x_0, t, noise=noises[0]), x_0
    return diffuse_t(x_0, t, noise=noises[0]), diffuse_t(x_0, t_next, noise=noises[1])


# ------------------------------------------------------------------ loss (ref :382-445)
def _loss_kind(loss_func):
    name = loss_func if isinstance(loss_func, str) else getattr(loss_func, "__name__", None)
    if name is None:
        name = cfg.LOSS_FUNC
    if name not in LOSS_KINDS:
        raise NotImplementedError(f"loss function {name}")
    return LOSS_KINDS[name]


def loss(model: Denoiser, x_t, x_1, x_tgt, x_0, image_clip, text_clip, mask, idx, loss_func=None, *, cfg_uniform=None):
    S, B, L, C_ = cfg.SAMPLE_SIZE, cfg.BATCH_SIZE, cfg.MAX_LENGTH, cfg.IN_CHANNEL
    assert x_t.shape == (S * B, L, C_)
    assert x_1.shape == x_0.shape == (B, L, C_)
    assert image_clip.shape == text_clip.shape == (B, 512)
    assert mask.shape == (B, L)
    assert idx.shape == (B, L)
    kind = _loss_kind(loss_func)
    if model.te:
        from . import train_embedding
        return train_embedding.loss(model, x_t, x_1, x_tgt, x_0, image_clip, text_clip, mask, idx, kind, cfg_uniform=cfg_uniform)
    dev = model.device
    lib = _lib.lib()
    Nt = S * B
    w = float(cfg.CLASSIFIER_FREE_WEIGHT)
    want_grad = torch.is_grad_enabled()

    # ---- classifier-free-guidance draw (ref :406-412): on the host, so that Ng is known without a device->host synchronisation
    gi = None
    Ng = 0
    if w > 0:
        draw = _state.pop("cfg_pending", None)          # train_func draws ahead (it places the x_1 rows behind the guided copies)
        if draw is None or draw["Nt"] != Nt or cfg_uniform is not None and not draw.get("injected"):
            draw = _guidance_draw(model, Nt, cfg_uniform)
        Ng = draw["Ng"]
        gi = draw["gi"][:Ng] if Ng else None
    model.params.text_unused = (not model.concat) and Ng == 0
    N = Nt + Ng + B
    # without a guided row the text row is masked as a key everywhere and its outputs are unused: leave it out (Tk = L+1)
    drop_txt = model.concat and Ng == 0 and cfg.DROP_UNUSED_TEXT_ROW
    # guidance: the number of guided copies is a fresh Binomial(Nt, 1 - p) draw every step -> one workspace sized for the worst case
    cap = (2 * Nt + B) if w > 0 else N
    ws = model._workspace(N, L, drop_txt, cap)
    Tk = ws["Tk"]

    # ---- loss scales (ref :77-87) and scratch
    sc = ws.get("loss_sc")
    if sc is None:
        sc = ws["loss_sc"] = dict(per_seq=torch.zeros(ws["cap"], dtype=torch.float32, device=dev),
                                  gscale=torch.zeros(ws["cap"], dtype=torch.float32, device=dev),
                                  ring=torch.zeros(LOSS_RING, 8, dtype=torch.float32, device=dev), slot=0)
    # the reference returns fresh tensors; here every call gets its own slot of a ring (no extra kernel), so losses a caller keeps
    # (a list of per-step values, an epoch accumulator) are not overwritten by the next step
    sc["slot"] = (sc["slot"] + 1) % LOSS_RING
    out = sc["ring"][sc["slot"]]
    inv768 = 1.0 / 768.0
    if kind == 0:
        sa, sb = inv768 / Nt, inv768 / B
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
    M = (Nt + B) * L
    cw = model._ce_workspace(M)
    dx = ws["dx_out"]

    # ---- one stacked encoder batch: [x_t rows | guided copies | x_1 rows]
    xin = ws["xin"]
    if x_t.data_ptr() != xin.data_ptr():                  # (train_func lets q_sample write into the workspace directly)
        xin[:Nt].copy_(x_t)
    if x_1.data_ptr() != xin[Nt + Ng:].data_ptr():
        xin[Nt + Ng:N].copy_(x_1)

    def resident(t_, dtype):
        return t_.device == dev and t_.dtype == dtype and t_.is_contiguous()
    fast = (not model.temb and resident(image_clip, torch.float32) and resident(text_clip, torch.float32)
            and resident(mask, torch.int64) and resident(idx, torch.int64))
    if fast and Ng:
        # guided step: the cats / repeats / index copies of ref :406-415, the guided rows' inputs and the zero fill of their dx rows are ONE kernel
        st0 = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.dic_cfg_prep(_p(image_clip), _p(text_clip), _p(mask), _p(idx), _p(gi), S, B, L, Tk, Ng, 768, _p(ws["img_in"]), _p(ws["txt_in"]),
                                    _p(ws["kmask"]), _p(ws["addtxt"]), _p(cw["tgt"]) if cfg.USE_PROB_LOSS else 0, _p(sc["gscale"]) if want_grad else 0,
                                    sa, sb, _p(xin), _p(dx) if want_grad else 0, st0), "cfg_prep")
        x_out = model.encode(xin[:N], None, None, None, drop_txt=drop_txt, cap=cap)
    elif fast:
        # no guidance: the repeats / hstacks / cats of ref :406-415, 426 and the target ids of :434-437 are ONE kernel writing the
        # encoder's and the rounding head's input buffers (no ATen kernel on the step)
        st0 = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.dic_step_prep(_p(image_clip), _p(text_clip), _p(mask), _p(idx), S, B, L, Tk, _p(ws["img_in"]),
                                     0 if ws["mode"] == 2 else _p(ws["txt_in"]), _p(ws["kmask"]), _p(ws["addtxt"]),
                                     _p(cw["tgt"]) if cfg.USE_PROB_LOSS else 0, _p(sc["gscale"]) if want_grad else 0, sa, sb, st0), "step_prep")
        x_out = model.encode(xin[:N], None, None, None, drop_txt=drop_txt, cap=cap)
    else:
        img = image_clip.to(dev, torch.float32)
        txt = text_clip.to(dev, torch.float32)
        img_rep, txt_rep = img.repeat(S, 1), txt.repeat(S, 1)
        m = (mask.to(dev) != 0).to(torch.uint8)
        m_rep = m.repeat(S, 1)
        if model.concat:
            one_t, one_b = torch.ones(Nt, 1, dtype=torch.uint8, device=dev), torch.ones(B, 1, dtype=torch.uint8, device=dev)
            plain_t = torch.cat([m_rep, one_t] if drop_txt else [m_rep, one_t, 0 * one_t], 1)
            plain_b = torch.cat([m, one_b] if drop_txt else [m, one_b, 0 * one_b], 1)
        else:
            plain_t, plain_b = m_rep, m
        add_txt = torch.zeros(N, dtype=torch.uint8, device=dev)
        if Ng:
            xin[Nt:Nt + Ng].copy_(x_t[gi])
            g_mask = torch.cat([m_rep[gi], one_t[:Ng], one_t[:Ng]], 1) if model.concat else m_rep[gi]
            ic = torch.cat([img_rep, img_rep[gi], img])
            tc = torch.cat([txt_rep, txt_rep[gi], txt])
            km = torch.cat([plain_t, g_mask, plain_b])
            add_txt[Nt:Nt + Ng] = 1
        else:
            ic, tc, km = torch.cat([img_rep, img]), torch.cat([txt_rep, txt]), torch.cat([plain_t, plain_b])
        tidx = None
        if model.temb:          # optional timestep embedding: row s*B+b carries t[s], a guided copy its source row's, the x_1 rows t = 1
            tv = getattr(model, "_step_t", None)
            assert tv is not None and tv.numel() == S, "cfg.TIMESTEP_EMBEDDING needs the step's t-vector (train_func passes it)"
            tt = tv.reshape(S, 1).to(dev, torch.int32).repeat(1, B).reshape(Nt)
            tidx = torch.cat([tt] + ([tt[gi]] if Ng else []) + [torch.ones(B, dtype=torch.int32, device=dev)])
        x_out = model.encode(xin[:N], ic, tc, km, add_txt, drop_txt=drop_txt, cap=cap, tidx=tidx)
    st = model.ops.stream
    row = Tk * 768
    if Ng:
        _lib.check(lib.dic_cfg_mix_fwd(_p(x_out), _p(x_out) + Nt * row * 4, _p(gi), Ng, row, w, st), "cfg_mix_fwd")

    # ---- embedding losses (ref :77-87, 418, 428) + compact rows for the rounding head
    if want_grad and not fast:
        sc["gscale"][:Nt].fill_(sa)
        sc["gscale"][Nt:Nt + B].fill_(sb)
        if Ng:
            dx[Nt:Nt + Ng].zero_()
    tgt_t, tgt_rows = (x_0, B) if cfg.X_0_PREDICTION else (x_tgt, Nt)
    if not cfg.X_0_PREDICTION:
        assert x_tgt.shape == x_t.shape
    tgt_t = tgt_t.contiguous()
    x_0c = x_0.contiguous()
    es = model.es
    xr_copy = not (cfg.USE_PROB_LOSS and model.head_centered)          # (a mean-centred head input is written by dic_head_center below instead)
    _lib.check(lib.dic_emb_loss(model.dt, kind, _p(x_out), _p(tgt_t), tgt_rows, _p(sc["per_seq"]), _p(dx) if want_grad else 0,
                                _p(sc["gscale"]), _p(cw["xr"]) if xr_copy else 0, Nt, L, Tk, 768, st), "emb_loss")
    off = (Nt + Ng) * row * 4
    _lib.check(lib.dic_emb_loss(model.dt, kind, _p(x_out) + off, _p(x_0c), B, _p(sc["per_seq"]) + Nt * 4, (_p(dx) + off) if want_grad else 0,
                                _p(sc["gscale"]) + Nt * 4, (_p(cw["xr"]) + Nt * L * 768 * es) if xr_copy else 0, B, L, Tk, 768, st), "emb_loss")
    _lib.check(lib.dic_seg_sum(_p(sc["per_seq"]), Nt + B, Nt, sa, sb, _p(out), 0, st), "seg_sum")
    model._last_total = out[3]
    if cfg.USE_PROB_LOSS:
        model.center_head_input(cw, _p(x_out), Nt, _p(x_out) + off, B, L, Tk)          # (bf16 engines: xr <- bf16(x - mean row), its logits -> the head's bias)

    # ---- rounding loss (ref :432-445): streaming GEMM + logsumexp + gather, logits never materialised
    if cfg.USE_PROB_LOSS:
        if not fast:
            ids = idx.to(dev, torch.int64)
            cw["tgt"][:Nt * L].copy_(ids.repeat(S, 1).reshape(-1))
            cw["tgt"][Nt * L:].copy_(ids.reshape(-1))
        fused_ce = want_grad and model.ce_fused
        if fused_ce:
            model.rounding_train(cw["xr"], M, cw["tgt"], cw)
        else:
            model.rounding(cw["xr"], M, cw["tgt"], cw)
        ca = (1.0 / Nt) if kind in (0, 2) else (1.0 / cfg.BATCH_SIZE)
        cb = (1.0 / B) if kind in (0, 2) else (1.0 / cfg.BATCH_SIZE)
        rw = float(cfg.ROUNDING_WEIGHT)
        _lib.check(lib.dic_seg_sum(_p(cw["nll"]), M, Nt * L, rw * ca, rw * cb, _p(out) + 4 * 4, _p(out) + 2 * 4, st), "seg_sum")
        model._last_total = out[7]                       # x_t_loss + x_1_loss + prob_loss, summed by the kernel (ref :481)
        if want_grad:
            dxr = model.rounding_backward(cw, M, Nt * L, rw * ca, rw * cb)
            if fused_ce:                  # dxr lacks the per-row factor row_scale / Z (engine.rounding_train): applied while it is added
                _lib.check(lib.dic_add_rows_scaled(_p(dx), _p(dxr), _p(cw["inv_z"]), rw * ca, Nt, L, Tk, 768, st), "add_rows_scaled")
                _lib.check(lib.dic_add_rows_scaled(_p(dx) + off, _p(dxr) + Nt * L * 768 * 4, _p(cw["inv_z"]) + Nt * L * 4, rw * cb, B, L, Tk, 768, st),
                           "add_rows_scaled")
            else:
                _lib.check(lib.dic_add_rows(_p(dx), _p(dxr), Nt, L, Tk, 768, st), "add_rows")
                _lib.check(lib.dic_add_rows(_p(dx) + off, _p(dxr) + Nt * L * 768 * 4, B, L, Tk, 768, st), "add_rows")
        prob = out[6]
    else:
        prob = torch.zeros((), dtype=torch.float32, device=dev)
    if want_grad and Ng:
        _lib.check(lib.dic_cfg_mix_bwd(_p(dx), _p(dx) + Nt * row * 4, _p(gi), Ng, row, w, st), "cfg_mix_bwd")
    model._pending = want_grad
    return out[0], out[1], prob


from .options import OPT


# ------------------------------------------------------------------ AdamW (ref :335)
class AdamW:
    """torch.optim.AdamW semantics (lr, betas=(0.9,0.999), eps=1e-8, weight_decay=0.01, one param group covering
    every tensor) as ONE fused HIP launch over the flat parameter buffer.  `param_groups[0]['lr']` is mutable, as the
    reference's epoch loop expects (ref :520-522)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        params = list(params)
        store = getattr(params[0], "_dic_store", None)
        if store is None:
            raise TypeError("dic.AdamW needs the tensors returned by Denoiser.parameters()")
        self.store = store
        self.param_groups = [dict(params=params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)]
        self.m = torch.zeros_like(store.P)
        self.v = torch.zeros_like(store.P)
        self.t = 0
        self.grad_scale = 1.0          # 1/world_size after the RCCL sum (parallel.py)

    def zero_grad(self, set_to_none=False):
        # Denoiser.backward overwrites every gradient it produces; only the few slots a step may leave untouched (position rows
        # beyond the sequence, text_linear when the text row is skipped) need clearing, and backward() does that itself when
        # this flag is set -- instead of a 347 MB fill per step.
        self.store.zero_pending = True

    def _launch(self, lo, hi):
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        s = self.store
        st = torch.cuda.current_stream().cuda_stream
        sh = (_p(s.Pb) + 2 * lo) if s.Pb is not None else 0
        sl = (_p(s.Pl) + 2 * lo) if s.Pl is not None else 0          # split weights: the low-order bf16 half, refreshed in the same pass
        _lib.check(_lib.lib().dic_adamw_hl(_p(s.P) + 4 * lo, _p(s.G) + 4 * lo, _p(self.m) + 4 * lo, _p(self.v) + 4 * lo, sh, sl, hi - lo,
                                           float(g["lr"]), b1, b2, g["eps"], g["weight_decay"], 1.0 - b1 ** self.t, 1.0 - b2 ** self.t,
                                           self.grad_scale, st), "adamw")

    # Streamed stepping (single-GPU training): the update of an encoder layer's slice is an HBM-bound pass that can run on the
    # weight-gradient stream as soon as that layer's gradients are final, under the MFMA-bound backward of the layers below it.
    # begin_step() ... step_range(lo, hi) per finished slice ... step() then only covers what is left.  Same arithmetic, same
    # step counter: the result is bit-identical to one launch over the whole buffer.
    def begin_step(self):
        self.t += 1
        self._streamed = []

    def step_range(self, lo, hi):
        self._launch(lo, hi)
        self._streamed.append((lo, hi))

    def step(self):
        streamed = getattr(self, "_streamed", None)
        self._streamed = None
        if streamed is None:
            self.t += 1
            streamed = []
        # torch skips tensors whose .grad is None (unused this step: text_linear under "add" fusion without guidance)
        for lo, hi in self.store.active_ranges():
            cur = lo
            for a, b in sorted(streamed):
                if b <= cur or a >= hi:
                    continue
                if a > cur:
                    self._launch(cur, a)
                cur = max(cur, b)
            if cur < hi:
                self._launch(cur, hi)

    def state_dict(self):
        return dict(t=self.t, m=self.m.clone(), v=self.v.clone(), param_groups=[{k: v for k, v in self.param_groups[0].items() if k != "params"}])

    def load_state_dict(self, sd):
        self.t = sd["t"]
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.param_groups[0].update(sd["param_groups"][0])


# ------------------------------------------------------------------ training step (ref :458-486)
def train_func(model: Denoiser, trainer, x, train=True, *, t=None, noises=None, cfg_uniform=None):
    """ref :458-486.  Returns (l, x_t_loss, x_1_loss, prob_loss) as 0-dim device tensors.  LIFETIME: they are views into a ring of
    LOSS_RING (4096) result slots per encoder workspace, written by the loss kernels themselves (no ATen kernel on the step): a value stays
    valid for at least the next LOSS_RING // 2 calls (4095 unless a graph.GraphedTrainStep re-captures in between and restarts at the ring's head).  Accumulating (`acc += l`, what the reference's epoch loop does, ref :530-533) or reading them is always
    fine; a caller that keeps per-step tensors for longer than that must `.clone()` them."""
    from . import parallel
    dev = model.device
    x_0 = model.embedding(x["input_ids"].to(dev))
    S = cfg.SAMPLE_SIZE
    if t is None:
        t = _draw_t(S, dev)                               # one t-vector per step shared by the (global) batch (ref :461)
    t = t.to(dev)
    nz = list(noises) if noises is not None else [None, None, None]
    # without classifier-free guidance the stacked encoder batch is [x_t rows | x_1 rows]: q_sample writes straight into it
    out_t = out_1 = None
    if cfg.X_0_PREDICTION and not model.te:
        Bc, Lc = x_0.shape[0], x_0.shape[1]
        Nt = t.numel() * Bc
        if float(cfg.CLASSIFIER_FREE_WEIGHT) <= 0:
            xin = model._workspace(Nt + Bc, Lc, model.concat and cfg.DROP_UNUSED_TEXT_ROW)["xin"]
            out_t, out_1 = xin[:Nt], xin[Nt:]
        else:
            # guidance: the guided copies sit between the x_t and the x_1 rows, so their number is drawn now (host side, no sync)
            draw = _guidance_draw(model, Nt, cfg_uniform)
            draw["injected"] = cfg_uniform is not None
            _state["cfg_pending"] = draw
            Ng_ = draw["Ng"]
            xin = model._workspace(Nt + Ng_ + Bc, Lc, model.concat and cfg.DROP_UNUSED_TEXT_ROW and Ng_ == 0, 2 * Nt + Bc)["xin"]
            out_t, out_1 = xin[:Nt], xin[Nt + Ng_:Nt + Ng_ + Bc]
    if cfg.X_0_PREDICTION:
        x_t = diffuse_t(x_0, t, noise=nz.pop(0), out=out_t)
        x_tgt = None
    else:
        t_next = torch.max(t - cfg.X_T_STEP_INTERVAL, torch.zeros_like(t))
        x_t, x_tgt = generate_diffuse_pair(x_0, t, t_next, noises=(nz.pop(0), nz.pop(0)))
    x_1 = diffuse_t(x_0, _t_one(dev), noise=nz.pop(0), out=out_1)
    model._step_t = t
    if train:
        trainer.zero_grad()
        if not isinstance(trainer, AdamW):
            # a torch optimizer's zero_grad() only drops the .grad views: the slots this backward will not write (position rows
            # beyond the sequence, text_linear when the text row is skipped) must still be cleared in the flat buffer
            model.params.zero_pending = True
    model._last_total = None
    x_t_loss, x_1_loss, prob_loss = loss(model, x_t, x_1, x_tgt, x_0, x["image_clip"], x["text_clip"], x["attention_mask"],
                                         x["input_ids"], cfg.LOSS_FUNC, cfg_uniform=cfg_uniform)
    # l = x_t_loss + x_1_loss + prob_loss (ref :481): the loss kernels already wrote that sum next to the three terms
    l = model._last_total if model._last_total is not None else x_t_loss + x_1_loss + prob_loss
    if train:
        if not model._pending:
            raise RuntimeError("train_func(train=True) called under torch.no_grad()")
        reducer = parallel.GradReducer(model)
        layer_done = reducer.layer_done if reducer.active else None
        if layer_done is None and isinstance(trainer, AdamW) and OPT.streamed_adamw:
            store = model.params
            trainer.begin_step()

            def layer_done(i):        # runs on the weight-gradient stream, ordered after layer i's last gradient kernel
                lo = store.off(f"L{i}.Wqkv")
                hi = store.off(f"L{i + 1}.Wqkv") if i + 1 < store.n_layers else store.off("pos")
                trainer.step_range(lo, hi)
        try:
            model.backward(layer_done=layer_done)
            if model.te:
                from . import train_embedding
                train_embedding.backward_tail(model, t, None if cfg.X_0_PREDICTION else t_next)   # projections, q_sample, embedding (x_0 carries gradient)
            reducer.finish(trainer)
        finally:
            model.ops.default_cu_cap = 0          # (DIC_DP_CU_CAP: a failed backward / collective must not leave validation and sampling capped)
        if not isinstance(trainer, AdamW):
            model.params.relink_grads()
        trainer.step()
        if not isinstance(trainer, AdamW):
            model.refresh_shadows()
    return l, x_t_loss, x_1_loss, prob_loss


# ------------------------------------------------------------------ validation (ref :488-501)
def set_loaders(val_loader=None, trainer=None):
    _state["val_loader"], _state["trainer"] = val_loader, trainer


def validate(model: Denoiser, val_loader=None):
    val_loader = val_loader if val_loader is not None else _state["val_loader"]
    acc = [0, 0, 0]
    model.eval()
    with torch.no_grad():
        n = 0
        for x in val_loader:
            _, a, b, c = train_func(model, _state["trainer"], x, train=False)
            acc = [acc[0] + a.clone(), acc[1] + b.clone(), acc[2] + c.clone()]
            n += 1
    model.check_ids(sync=True)          # an out-of-range token id in the last batch would otherwise never surface (nn.Embedding raises at once)
    model.train()
    return acc[0] / n, acc[1] / n, acc[2] / n


# ------------------------------------------------------------------ sampling loop (ref :611-621, COCO_BLEU.py:249-256)


@torch.no_grad()
def sample(model: Denoiser, image_clip, steps=5, *, start=None, return_hidden=False):
    """x_0-prediction refinement from pure noise: `steps` encoder passes feeding the prediction straight back in,
    then round to token ids.  The logits are only needed after the last pass, and only their argmax: the
    rounding GEMM runs in exact fp32 (MFMA f32) with a streaming arg-max, so ids match the CPU oracle bit-for-bit.

    What is constant over the loop is done once: the CLIP rows and the key mask are packed into the encoder's workspace and the CLIP
    projections computed by the first pass only; every pass reads rows [:, :L] of the previous pass's x_out in place (strided read in
    dic_fuse_ln_fwd_x) instead of compacting them; from the third pass on the passes are launch-for-launch identical and are replayed as
    one hipGraph.  With cfg.TIMESTEP_EMBEDDING (not in the reference, which has no timestep input): the first pass sees pure noise and
    takes the table row of t = STEP_TOT - 1, every later pass sees an x_0 estimate and takes row 1, the row training uses for the x_1 pass."""
    if model.te:
        from . import train_embedding
        return train_embedding.sample(model, image_clip, steps, start, return_hidden)
    dev = model.device
    B, L = image_clip.shape[0], cfg.MAX_LENGTH
    img = image_clip.to(dev, torch.float32).reshape(B, 512)
    restored = (start.to(dev, torch.float32) if start is not None else torch.randn(B, L + 2, cfg.IN_CHANNEL, device=dev)).contiguous()
    assert restored.shape[0] == B and restored.shape[1] >= L and restored.shape[2] == 768
    # the text row is masked as a key ([1, 0]) and only rows < L are fed back: skip it unless the caller wants the full hidden state
    drop_txt = model.concat and cfg.DROP_UNUSED_TEXT_ROW and not return_hidden
    ws = model._workspace(B, L, drop_txt)
    Tk = ws["Tk"]
    ws["img_in"][:B].copy_(img)
    ws["txt_in"][:B].zero_()
    ws["kmask"][:B].fill_(1)
    if model.concat and not drop_txt:
        ws["kmask"][:B, L + 1].zero_()
    ws["addtxt"][:B].zero_()
    t_first = t_later = None
    if model.temb:
        t_first = torch.full((B,), int(cfg.STEP_TOT) - 1, dtype=torch.int32, device=dev)
        t_later = torch.ones(B, dtype=torch.int32, device=dev)
    x_view = (restored.data_ptr(), restored.shape[1] * 768, B, L)
    x_out, graph = None, None
    if steps <= 0:                                            # no refinement pass: round the start tensor itself (the reference's loop body never runs)
        x = restored[:, :L, :].contiguous()
        _, ids, _ = model.rounding(x.reshape(B * L, 768), B * L, dtype=_lib.DIC_F32)
        ids = ids.clone().reshape(B, L)
        return (ids, restored.clone()) if return_hidden else ids
    # a replayed pass re-uses the captured dropout seed: only capture when no mask is drawn (eval mode, or p = 0)
    no_dropout = (not model.training) or (model.p_hidden == 0.0 and model.p_attn == 0.0)
    # a forward-only pass is one stream of kernels: its GEMMs may finish their last round with shorter tiles (include/dic_hip.h,
    # dic_gemm_set_two_heights; 7.56 -> 7.47 ms per pass at B = 2048 -- the training step keeps the switch off)
    lib = _lib.lib()
    prev_two = lib.dic_gemm_set_two_heights(1 if (OPT.sample_two_heights or OPT.gemm_two_heights) else 0)
    # ... and its k-contiguous GEMMs without dropout (QKV, out-proj + residual, FFN-2 + residual, the MLM-head transform) run on the hand-scheduled
    # four-wave kernel where their shape allows (include/dic_hip.h, dic_gemm_set_w4a: token count a multiple of 256): 8.06 -> 7.90 ms per pass at
    # B = 2048 (profiles/r04_sampling_w4a_ab.txt).  The training step keeps it off: at the 1400 W package limit its denser MFMA stream is paid back
    # in clock (profiles/r04_power_probe.txt)
    prev_w4a = lib.dic_gemm_set_w4a(1 if (OPT.sample_w4a or OPT.gemm_w4a) else 0)
    try:
        for k in range(steps):
            if graph is not None:
                graph.replay()
                continue
            if k >= 1:
                x_view = (x_out.data_ptr(), Tk * 768, B, L)
            # (raw: the passes run without the parity mode's mean-row corrections -- a sampling pass feeds a per-row argmax, there is no batch mean
            # whose row-common roundings would need protecting, and the 37 small launches per pass would cost 5 %; options.sample_raw)
            run = lambda: model.encode(None, None, None, None, drop_txt=drop_txt, x_view=x_view, inputs_ready=k > 0, tidx=t_first if k == 0 else t_later,
                                       raw=OPT.sample_raw)
            if OPT.sample_graph and no_dropout and k == 2 and steps >= 8:
                seed_before = model._seed
                try:
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        run()
                    graph.replay()
                    continue
                except Exception as e:                        # capture not possible here: run the loop launch by launch, and say so
                    import warnings
                    warnings.warn(f"sample(): hipGraph capture of a denoising pass failed ({type(e).__name__}: {e}); running launch by launch")
                    graph = None
                    model._seed = seed_before
            x_out = run()
    finally:
        lib.dic_gemm_set_two_heights(prev_two)
        lib.dic_gemm_set_w4a(prev_w4a)
    x = x_out[:, :L, :].contiguous()
    _, ids, _ = model.rounding(x.reshape(B * L, 768), B * L, dtype=_lib.DIC_F32)
    ids = ids.clone().reshape(B, L)
    return (ids, x_out.clone()) if return_hidden else ids


def dedup_columns(ids: torch.Tensor) -> torch.Tensor:
    """`indexes.unique_consecutive(dim=-1)` (ref :621): removes a COLUMN only when it repeats the previous column
    across the whole batch (per-token de-duplication only when B == 1)."""
    return ids.unique_consecutive(dim=-1)

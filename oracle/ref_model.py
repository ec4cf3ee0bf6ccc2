"""CPU oracle: a plain fp32 PyTorch restatement of the reference hot path.

TEST INFRASTRUCTURE.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this file; the product path (`diffusion-image-captioning_amd/`) never does
and fails loudly when its HIP library is missing.

Parity status: PINNED -- `tests/test_oracle_golden.py` checks every function here against the
vectors in `tests/golden/`, which `oracle/gen_golden.py` produced by executing the reference's own
definitions (AST-extracted from /root/reference/CLIP-DDPM.py) on top of HuggingFace DistilBERT
(transformers 5.15.0, unpinned by the reference) in the build container.

Every function cites the reference lines it restates.  `hf:` = transformers/models/distilbert/
modeling_distilbert.py (5.15.0).  The encoder is written out here instead of calling HF so the
oracle travels to the GPU box without `transformers`.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F


@dataclass
class Config:
    """The hyper-parameter block CLIP-DDPM.py:55-114 as a record (reference: module globals)."""
    BATCH_SIZE: int = 8
    MAX_LENGTH: int = 16
    SAMPLE_SIZE: int = 100
    STEP_TOT: int = 1000
    COSIN_SCHEDULE: bool = True
    BETA_MIN: float = 0.0001
    BETA_MAX: float = 0.02
    ROUNDING_WEIGHT: float = 0.5
    LOSS_FUNC: str = "series_sum_sample_mean"
    CLIP_ADDING_METHOD: str = "concat"
    CLASSIFIER_FREE_WEIGHT: float = 0.0
    CLASSIFIER_FREE_PROB: float = 0.2
    X_0_PREDICTION: bool = True
    X_T_STEP_INTERVAL: int = 100
    USE_X_T_LOSS: bool = True
    USE_X_1_LOSS: bool = True
    USE_PROB_LOSS: bool = True
    IN_CHANNEL: int = 768
    TRAIN_EMBEDDING: bool = False  # :98-102: learned 16-d embedding + projections instead of the frozen DistilBERT one
    LEARNING_RATE: float = 1e-4
    # denoiser (HF DistilBertConfig defaults, CLIP-DDPM.py:236,330)
    n_layers: int = 6
    n_heads: int = 12
    dim: int = 768
    hidden_dim: int = 3072
    vocab: int = 30522
    dropout: float = 0.0          # parity runs use p=0; the reference default is 0.1
    attention_dropout: float = 0.0


# ------------------------------------------------------------------ schedule  (CLIP-DDPM.py:337-346)
def alpha_cumprod(cfg: Config) -> torch.Tensor:
    if cfg.COSIN_SCHEDULE:
        s = 0.008

        def sched(t):
            return torch.cos(math.pi / 2 * (t / cfg.STEP_TOT + s) / (1 + s)) ** 2
        ts = torch.arange(cfg.STEP_TOT)
        return sched(ts) / sched(torch.zeros(1))
    betas = torch.hstack([torch.zeros(1), torch.linspace(cfg.BETA_MIN, cfg.BETA_MAX, cfg.STEP_TOT)])
    return torch.cumprod((1 - betas)[:-1], 0)


# ------------------------------------------------------------------ q_sample  (CLIP-DDPM.py:347-362)
def diffuse_t(x: torch.Tensor, t: torch.Tensor, noise: torch.Tensor, ac: torch.Tensor) -> torch.Tensor:
    """x [B,L,C], t [S,...] int64, ONE noise tensor [B,L,C] shared by all S -> [S*B, L, C] (s-major)."""
    b, l, c = x.shape
    shp = (t.numel(), 1, 1, 1)
    mean = torch.sqrt(ac[t].reshape(shp)) * x
    eps = noise * torch.sqrt(1 - ac[t]).reshape(shp)
    return (mean + eps).reshape(t.numel() * b, l, c)


# ------------------------------------------------------------------ embedding losses (CLIP-DDPM.py:77-87)
def series_sum_sample_mean(x_hat, x, cfg):
    return (x_hat - x).abs().sum(dim=1).mean()


def series_sum(x_hat, x, cfg):
    return (x_hat - x).abs().sum() / cfg.BATCH_SIZE / 768 / 100


def mse_series_mean(x_hat, x, cfg):
    return ((x_hat - x) ** 2).sum(dim=[-2, -1]).sqrt().mean()


def mse_series_sum(x_hat, x, cfg):
    return ((x_hat - x) ** 2).sum(dim=[-2, -1]).sqrt().sum() / cfg.BATCH_SIZE


LOSS_FUNCS = {f.__name__: f for f in (series_sum_sample_mean, series_sum, mse_series_mean, mse_series_sum)}


# ------------------------------------------------------------------ per-op references (also used by the GPU op tests)
def layer_norm(x, w, b, eps=1e-12):
    """hf:100,116 / hf:236,253 / hf:439,512 -- nn.LayerNorm(dim, eps=1e-12)."""
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def gelu(x):
    """hf `get_activation("gelu")` = exact erf GELU (hf:223, 511)."""
    return F.gelu(x)


def attention(q, k, v, key_mask, n_heads, drop_mask=None, p_drop=0.0):
    """hf:136-147 eager attention.  q,k,v [N,T,D]; key_mask [N,T] (1 = attend).  softmax(QK^T/sqrt(dh)
    + (-inf on masked keys)) V, heads split as view(N,T,H,dh).transpose(1,2) (hf:183-185)."""
    n, t, d = q.shape
    dh = d // n_heads
    qh = q.view(n, t, n_heads, dh).transpose(1, 2)
    kh = k.view(n, t, n_heads, dh).transpose(1, 2)
    vh = v.view(n, t, n_heads, dh).transpose(1, 2)
    s = torch.matmul(qh, kh.transpose(2, 3)) * (dh ** -0.5)
    s = s.masked_fill(key_mask[:, None, None, :] == 0, float("-inf"))
    p = F.softmax(s, dim=-1)
    if drop_mask is not None:
        p = p * drop_mask / (1.0 - p_drop)
    o = torch.matmul(p, vh)
    return o.transpose(1, 2).reshape(n, t, d)


class Denoiser:
    """`class DistilBertModel` (CLIP-DDPM.py:227-323) + the HF encoder it wraps, as explicit tensor math.

    `params` is a dict name -> tensor using the reference module tree's names
    (`model.distilbert.transformer.layer.0.attention.q_lin.weight`, `image_linear.weight`, ...).
    `embedding` / `lm_weight` are the frozen token embedding and rounding head (bias zeroed, :245-247).
    """

    def __init__(self, cfg: Config, params: dict, embedding: torch.Tensor, lm_weight: torch.Tensor):
        self.cfg = cfg
        self.p = params
        self.E = embedding
        self.W_lm = lm_weight

    def parameters(self):
        """Order of CLIP-DDPM.py:258-269."""
        names = [n for n in self.p]
        return [self.p[n] for n in names]

    def embedding(self, ids):            # CLIP-DDPM.py:459
        return (self.p["embedding.weight"] if self.cfg.TRAIN_EMBEDDING else self.E)[ids]

    def lm_head(self, h):                # CLIP-DDPM.py:323 (bias == 0; TRAIN_EMBEDDING: trainable, bias=False, :240)
        return h @ (self.p["lm_head.weight"] if self.cfg.TRAIN_EMBEDDING else self.W_lm).t()

    # ---- HF encoder + MLM-head transform (hf:92-118, 150-259, 501-513)
    def encoder(self, x, key_mask):
        p, cfg = self.p, self.cfg
        t = x.shape[1]
        pre = "model.distilbert."
        h = x + p[pre + "embeddings.position_embeddings.weight"][:t]
        h = layer_norm(h, p[pre + "embeddings.LayerNorm.weight"], p[pre + "embeddings.LayerNorm.bias"])
        for i in range(cfg.n_layers):
            lp = pre + f"transformer.layer.{i}."
            q = F.linear(h, p[lp + "attention.q_lin.weight"], p[lp + "attention.q_lin.bias"])
            k = F.linear(h, p[lp + "attention.k_lin.weight"], p[lp + "attention.k_lin.bias"])
            v = F.linear(h, p[lp + "attention.v_lin.weight"], p[lp + "attention.v_lin.bias"])
            ctx = attention(q, k, v, key_mask, cfg.n_heads)
            a = F.linear(ctx, p[lp + "attention.out_lin.weight"], p[lp + "attention.out_lin.bias"])
            sa = layer_norm(a + h, p[lp + "sa_layer_norm.weight"], p[lp + "sa_layer_norm.bias"])
            f = gelu(F.linear(sa, p[lp + "ffn.lin1.weight"], p[lp + "ffn.lin1.bias"]))
            f = F.linear(f, p[lp + "ffn.lin2.weight"], p[lp + "ffn.lin2.bias"])
            h = layer_norm(f + sa, p[lp + "output_layer_norm.weight"], p[lp + "output_layer_norm.bias"])
        u = F.linear(h, p["model.vocab_transform.weight"], p["model.vocab_transform.bias"])
        return layer_norm(gelu(u), p["model.vocab_layer_norm.weight"], p["model.vocab_layer_norm.bias"])

    # ---- wrapper forward (CLIP-DDPM.py:271-323)
    def forward(self, x, image_clip, text_clip, mask, concat_mask, with_logits=True):
        cfg, p = self.cfg, self.p
        n = x.shape[0]
        L = cfg.MAX_LENGTH
        assert x.shape == (n, L, cfg.IN_CHANNEL)
        assert image_clip.shape == text_clip.shape == (n, 1, 512)
        assert mask.shape == (n, L) and concat_mask.shape == (n, 2)
        guided = concat_mask[:, 1] == 1
        if cfg.TRAIN_EMBEDDING:            # :292-293
            x = F.linear(x, p["input_projection.weight"], p["input_projection.bias"])
        img = F.linear(image_clip, p["image_linear.weight"], p["image_linear.bias"])
        txt = F.linear(text_clip, p["text_linear.weight"], p["text_linear.bias"])
        mask = mask.to(torch.int64)
        if cfg.CLIP_ADDING_METHOD == "concat":
            one = torch.ones(n, 1, dtype=torch.int64)
            guided_mask = torch.hstack([mask, one, one])
            plain_mask = torch.hstack([mask, one, 0 * one])
            seg = p["segment_embedding.weight"][torch.tensor([0] * L + [1] * 2)]
            xg = xp = torch.hstack([x, img, txt]) + seg
        elif cfg.CLIP_ADDING_METHOD == "add":
            guided_mask = plain_mask = mask
            xp = x + img
            xg = xp + txt
        else:
            raise NotImplementedError(cfg.CLIP_ADDING_METHOD)
        x_out = self.encoder(xp, plain_mask)
        if cfg.CLASSIFIER_FREE_WEIGHT > 0 and guided.sum() != 0:
            w = cfg.CLASSIFIER_FREE_WEIGHT
            g_out = self.encoder(xg[guided], guided_mask[guided])
            mixed = (1 + w) * g_out - w * x_out[guided]
            x_out = x_out.index_put((guided.nonzero().squeeze(1),), mixed)
        if cfg.TRAIN_EMBEDDING:            # :319-320
            x_out = F.linear(x_out, p["output_projection.weight"], p["output_projection.bias"])
        logits = self.lm_head(x_out[:, :L, :]) if with_logits else None
        return logits, x_out

    __call__ = forward


# ------------------------------------------------------------------ loss (CLIP-DDPM.py:382-445)
def concat_mask_for(cfg: Config, n: int, cfg_uniform: torch.Tensor | None):
    """:406-412.  `cfg_uniform` is the U[0,1) draw of shape [n,1] (injected for parity)."""
    if cfg.CLASSIFIER_FREE_WEIGHT > 0:
        cm = (cfg_uniform > cfg.CLASSIFIER_FREE_PROB).float()
        cm[0] = 0
        cm[1] = 1
        return torch.hstack([torch.ones(n, 1), cm])
    return torch.tensor([1, 0]).repeat((n, 1))


def rounding_nll(logits, idx, cfg: Config):
    """:434-440: -log softmax gathered at the true ids, summed over seq; mean or sum/B by LOSS_FUNC."""
    nll = -(F.log_softmax(logits, dim=-1).gather(-1, idx.unsqueeze(-1))).squeeze(-1)
    if cfg.LOSS_FUNC in ("series_sum_sample_mean", "mse_series_mean"):
        return nll.sum(dim=1).mean()
    return nll.sum() / cfg.BATCH_SIZE


def loss(model: Denoiser, x_t, x_1, x_tgt, x_0, image_clip, text_clip, mask, idx, cfg_uniform=None):
    cfg = model.cfg
    S, B, L = cfg.SAMPLE_SIZE, cfg.BATCH_SIZE, cfg.MAX_LENGTH
    assert x_t.shape == (S * B, L, cfg.IN_CHANNEL)
    assert x_1.shape == x_0.shape == (B, L, cfg.IN_CHANNEL)
    lf = LOSS_FUNCS[cfg.LOSS_FUNC]
    rep = (S, 1, 1)
    image_clip = image_clip.unsqueeze(1)
    text_clip = text_clip.unsqueeze(1)
    cm = concat_mask_for(cfg, S * B, cfg_uniform)
    x_t_prob, x_t_hidden = model(x_t, image_clip.repeat(rep), text_clip.repeat(rep), mask.repeat((S, 1)), cm)
    if cfg.USE_X_T_LOSS:
        tgt = x_0.repeat(rep) if cfg.X_0_PREDICTION else x_tgt
        x_t_loss = lf(x_t_hidden[:, :L, :], tgt, cfg)
    else:
        x_t_loss = torch.zeros(())
    x_1_prob, x_1_hidden = model(x_1, image_clip, text_clip, mask, torch.tensor([1, 0]).repeat((B, 1)))
    x_1_loss = lf(x_1_hidden[:, :L, :], x_0, cfg) if cfg.USE_X_1_LOSS else torch.zeros(())
    if cfg.USE_PROB_LOSS:
        pl = rounding_nll(x_t_prob, idx.repeat((S, 1)), cfg) + rounding_nll(x_1_prob, idx, cfg)
    else:
        pl = torch.zeros(())
    return x_t_loss, x_1_loss, cfg.ROUNDING_WEIGHT * pl


# ------------------------------------------------------------------ AdamW (CLIP-DDPM.py:335: torch defaults)
class AdamW:
    """torch.optim.AdamW(lr, betas=(0.9,0.999), eps=1e-8, weight_decay=0.01) on every tensor, restated."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        self.params = list(params)
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)]
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self):
        g = self.param_groups[0]
        lr, (b1, b2), eps, wd = g["lr"], g["betas"], g["eps"], g["weight_decay"]
        self.t += 1
        bc1 = 1 - b1 ** self.t
        bc2 = 1 - b2 ** self.t
        for p, m, v in zip(self.params, self.m, self.v):
            if p.grad is None:
                continue
            p.mul_(1 - lr * wd)
            m.mul_(b1).add_(p.grad, alpha=1 - b1)
            v.mul_(b2).addcmul_(p.grad, p.grad, value=1 - b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
            p.addcdiv_(m, denom, value=-lr / bc1)


# ------------------------------------------------------------------ train step (CLIP-DDPM.py:458-486)
def train_func(model: Denoiser, trainer, x: dict, train=True, *, t, noises, cfg_uniform=None, ac=None):
    """`t` [S,1,1] int64 and `noises` (list of [B,L,C] tensors consumed in the order the reference draws
    them: x_t, (x_tgt when not X_0_PREDICTION), x_1) replace torch.randint / torch.normal."""
    cfg = model.cfg
    ac = alpha_cumprod(cfg) if ac is None else ac
    x_0 = model.embedding(x["input_ids"])
    noises = list(noises)
    x_t = diffuse_t(x_0, t, noises.pop(0), ac)
    x_tgt = None
    if not cfg.X_0_PREDICTION:                                     # :467
        t_next = torch.max(t - cfg.X_T_STEP_INTERVAL, torch.zeros_like(t))
        x_tgt = diffuse_t(x_0, t_next, noises.pop(0), ac)
    x_1 = diffuse_t(x_0, torch.ones(1, dtype=torch.int64), noises.pop(0), ac)
    if train:
        trainer.zero_grad()
    a, b, c = loss(model, x_t, x_1, x_tgt, x_0, x["image_clip"], x["text_clip"], x["attention_mask"],
                   x["input_ids"], cfg_uniform)
    l = a + b + c
    if train:
        l.backward()
        trainer.step()
    return l, a, b, c


# ------------------------------------------------------------------ sampling loop (CLIP-DDPM.py:611-621)
@torch.no_grad()
def sample(model: Denoiser, image_clip, steps=5, start=None):
    cfg = model.cfg
    b, L = image_clip.shape[0], cfg.MAX_LENGTH
    restored = start if start is not None else torch.randn(b, L + 2, cfg.IN_CHANNEL)
    for _ in range(steps):
        out, restored = model(restored[:, :L, :], image_clip.unsqueeze(1), torch.zeros_like(image_clip).unsqueeze(1),
                              torch.ones(b, L), torch.tensor([1, 0]).repeat(b, 1))
    return F.softmax(out, dim=-1).argmax(dim=-1), restored


def unique_consecutive_columns(ids: torch.Tensor) -> torch.Tensor:
    """`indexes.unique_consecutive(dim=-1)` (:621): drops a *column* only when it equals the previous
    column across the whole batch -- restated without the torch op."""
    keep = [0] + [j for j in range(1, ids.shape[1]) if not torch.equal(ids[:, j], ids[:, j - 1])]
    return ids[:, keep]


# ------------------------------------------------------------------ builders
def build(cfg: Config, state_np: dict, embedding_np, requires_grad=True) -> Denoiser:
    names = list(state_np)
    if cfg.CLIP_ADDING_METHOD != "concat":
        names = [n for n in names if n != "segment_embedding.weight"]
    params = {n: torch.from_numpy(state_np[n].copy()).requires_grad_(requires_grad) for n in names}
    E = torch.from_numpy(embedding_np)
    return Denoiser(cfg, params, E, E)

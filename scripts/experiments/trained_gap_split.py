#!/usr/bin/env python3
"""Localise the post-training rounding-loss gap (trained_gap_probe.py): the encoder outputs of the bf16w and the fp32 engine (same trained weights,
same batch) through the bf16 head and through the exact fp32 head -- which half of the arithmetic carries the 3e-4?"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dic = importlib.import_module("diffusion-image-captioning_amd")
_lib = importlib.import_module("diffusion-image-captioning_amd._lib")
B, S, L, NL = 512, 1, 16, 12
dic.cfg.update(BATCH_SIZE=B, SAMPLE_SIZE=S, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, VOCAB_SIZE=30522, CLASSIFIER_FREE_WEIGHT=0.0,
               CLIP_ADDING_METHOD="concat", LOSS_FUNC="series_sum_sample_mean", X_0_PREDICTION=True, ROUNDING_WEIGHT=0.5)
dev = "cuda:0"
E = torch.from_numpy(dic.synth.vocab_embedding(30522, 768, 0))
x = {k: torch.from_numpy(v).to(dev) for k, v in dic.synth.batch(B, L, 30522, seed=1).items()}
t = torch.from_numpy(dic.synth.timesteps(S, 100, 0))
nz = [torch.from_numpy(dic.synth.noise((B, L, 768), 3, f"eps{i}")) for i in range(2)]
kw = dict(config=dict(n_layers=NL, dropout=0.1, attention_dropout=0.1), device=dev, seed=0)
mw = dic.DistilBertModel(E, E, dtype="bf16w", **kw)
m32 = dic.DistilBertModel(E, E, dtype="fp32", **kw)
tr = dic.AdamW(mw.parameters(), lr=1e-4)
tgt = torch.cat([x["input_ids"].repeat(S, 1), x["input_ids"]]).reshape(-1).contiguous()
M = tgt.numel()
for nstep in (0, 200):
    for _ in range(nstep):
        dic.train_func(mw, tr, x)
    m32.load_state_dict(mw.state_dict())
    xo = {}
    for k, m in (("16", mw), ("32", m32)):
        m.eval()
        with torch.no_grad():
            dic.train_func(m, None, x, train=False, t=t, noises=nz)
        ws = m._saved
        xo[k] = ws["x_out"][:ws["N"], :L, :].reshape(M, 768).clone()
        m.train()
    d = (xo["16"] - xo["32"])
    print(f"after {nstep} steps: x_out bf16w vs fp32: rms diff {float(d.pow(2).mean().sqrt()):.3e} (rows rms {float(xo['32'].pow(2).mean().sqrt()):.3f}); "
          f"mean signed diff {float(d.mean()):+.2e}; mean row-norm ratio {float((xo['16'].norm(dim=1) / xo['32'].norm(dim=1)).mean()) - 1:+.2e}")
    res = {}
    for ek in ("16", "32"):
        for hk, hd in (("16", None), ("32", _lib.DIC_F32)):
            xr = xo[ek].to(torch.bfloat16).contiguous() if hd is None else xo[ek].contiguous()
            _, _, nll = mw.rounding(xr, M, tgt=tgt, dtype=hd)
            res[(ek, hk)] = float(nll.double().mean())
    base = res[("32", "32")]
    for k, v in res.items():
        print(f"   encoder {k[0]} -> head {k[1]}: mean nll {v:.6f}   relative to fp32/fp32 {(v - base) / base:+.2e}")
    # ---- what-if: mean-centred head input.  z = (x - xbar) W + xbar W: the row-common part xbar W in fp32 (a 768 x V matrix-vector product per step),
    # only the deviations go through bf16.  When the rows are nearly equal (an early-training denoiser predicts the mean) their bf16 rounding errors
    # are the SAME for every row and a batch mean does not average them out; the deviations' errors are independent again.
    import torch.nn.functional as F
    W32 = mw.W_lm[:30522]
    Wb = W32.to(torch.bfloat16).float()
    xs = xo["16"]

    def ce(logit_fn):
        tot = 0.0
        for a0 in range(0, M, 2048):
            lg = logit_fn(xs[a0:a0 + 2048])
            tot += float((torch.logsumexp(lg.double(), 1) - lg.double().gather(1, tgt[a0:a0 + 2048, None]).squeeze(1)).sum())
        return tot / M
    exact = ce(lambda v: v.double() @ W32.double().t())
    plain = ce(lambda v: v.to(torch.bfloat16).float() @ Wb.t())
    xbar = xs.mean(0, keepdim=True)
    cvec = (xbar.double() @ W32.double().t()).float()
    cent = ce(lambda v: (v - xbar).to(torch.bfloat16).float() @ Wb.t() + cvec)
    xb2 = torch.stack([xs[:S * B * L].mean(0), xs[S * B * L:].mean(0)])
    print(f"   torch emulation on the bf16w encoder's x_out: exact {exact:.6f}; bf16 operands {(plain - exact) / exact:+.2e}; mean-centred bf16 operands {(cent - exact) / exact:+.2e};"
          f" |xbar| {float(xbar.norm()):.2f}, rms |x - xbar| {float((xs - xbar).norm(dim=1).pow(2).mean().sqrt()):.3f}, block means differ by {float((xb2[0] - xb2[1]).norm()):.3f}")

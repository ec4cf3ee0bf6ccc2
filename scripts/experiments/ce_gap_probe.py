#!/usr/bin/env python3
"""Where the bf16 engine's rounding loss departs from the fp32 engine's at the bench shape (2.4e-4 relative, systematic).
Same weights / batch / noise / timesteps in both engines (dropout off); then, per token row of the rounding head:
  A  kernel nll (bf16 engine)            vs  fp64 CE on the SAME bf16 operands (xr bf16, W bf16)   -> error of the streaming CE kernel itself
  B  fp64 CE on bf16 operands            vs  fp64 CE on the unrounded operands (x_out fp32, W fp32) -> operand rounding
  C  fp64 CE on the bf16 engine's x_out  vs  fp64 CE on the fp32 engine's x_out (fp32 W both)      -> drift of the encoder output
reported as mean signed difference per token, its standard error, and relative to the mean nll."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dic = importlib.import_module("diffusion-image-captioning_amd")
B, S, L, NL = int(os.environ.get("B", "512")), 1, 16, int(os.environ.get("LAYERS", "12"))
dic.cfg.update(BATCH_SIZE=B, SAMPLE_SIZE=S, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, VOCAB_SIZE=30522, CLASSIFIER_FREE_WEIGHT=0.0,
               CLIP_ADDING_METHOD="concat", LOSS_FUNC="series_sum_sample_mean", X_0_PREDICTION=True, ROUNDING_WEIGHT=0.5)
dev = "cuda:0"
E = torch.from_numpy(dic.synth.vocab_embedding(30522, 768, 0))
x = {k: torch.from_numpy(v).to(dev) for k, v in dic.synth.batch(B, L, 30522, seed=1).items()}
t = torch.from_numpy(dic.synth.timesteps(S, 100, 0))
nz = [torch.from_numpy(dic.synth.noise((B, L, 768), 3, f"eps{i}")) for i in range(2)]
res = {}
state = None
for dt_ in ("bf16", "fp32"):
    m = dic.DistilBertModel(E, E, config=dict(n_layers=NL, dropout=0.1, attention_dropout=0.1), dtype=dt_, device=dev, seed=0)
    if state is None:
        state = m.state_dict()
    else:
        m.load_state_dict(state)
    m.eval()
    with torch.no_grad():
        r = dic.train_func(m, None, x, train=False, t=t, noises=nz)
    torch.cuda.synchronize()
    ws = m._saved
    M = (S * B + B) * L
    cw = m._ce_workspace(M)
    N = ws["N"]
    Tk = ws["Tk"]
    xo = ws["x_out"][:N, :L, :].reshape(M, 768).clone()       # rows in the order of xr: [x_t rows | x_1 rows] (no guided copies here)
    res[dt_] = dict(loss=[float(v) for v in r], x_out=xo, xr=cw["xr"].clone(), nll=cw["nll"].clone(), lse=cw["lse"].clone(), tgt=cw["tgt"].clone(),
                    W=m.W_lm[:30522].clone(), Wc=m.W_lm_c[:30522].clone())
    del m
    torch.cuda.empty_cache()


def ce64(xm, W, tgt, chunk=1024):
    out_nll, out_lse = [], []
    W64 = W.double()
    for i in range(0, xm.shape[0], chunk):
        lg = xm[i:i + chunk].double() @ W64.t()
        lse = torch.logsumexp(lg, 1)
        out_lse.append(lse)
        out_nll.append(lse - lg.gather(1, tgt[i:i + chunk, None]).squeeze(1))
    return torch.cat(out_nll), torch.cat(out_lse)


def report(name, a, b):
    d = (a.double() - b.double())
    n = d.numel()
    print(f"{name:78s} mean {d.mean().item():+.3e}  (s.e. {d.std().item() / n ** 0.5:.1e}, |max| {d.abs().max().item():.2e})  rel to mean nll {d.mean().item() / b.double().mean().item():+.2e}")


bf, f32 = res["bf16"], res["fp32"]
print("losses bf16:", bf["loss"], "\nlosses fp32:", f32["loss"])
print("rel diff   :", [abs(a - b) / abs(b) for a, b in zip(bf["loss"], f32["loss"])])
tgt = bf["tgt"]
nll_same, lse_same = ce64(bf["xr"], bf["Wc"], tgt)
report("A  bf16 kernel nll - fp64 CE on the same bf16 operands", bf["nll"], nll_same)
report("A' bf16 kernel lse - fp64 lse on the same bf16 operands", bf["lse"], lse_same)
nll_unr, _ = ce64(bf["x_out"], bf["W"], tgt)
report("B  fp64 CE(bf16 xr, bf16 W) - fp64 CE(fp32 x_out of the bf16 engine, fp32 W)", nll_same, nll_unr)
nll_xr_only, _ = ce64(bf["xr"], bf["W"], tgt)
report("B1   of which rounding xr only (W fp32)", nll_xr_only, nll_unr)
nll_w_only, _ = ce64(bf["x_out"], bf["Wc"], tgt)
report("B2   of which rounding W only (x fp32)", nll_w_only, nll_unr)
nll_f32, _ = ce64(f32["x_out"], f32["W"], tgt)
report("C  fp64 CE(x_out of bf16 engine) - fp64 CE(x_out of fp32 engine), fp32 W", nll_unr, nll_f32)
report("D  fp32 kernel nll - fp64 CE on its own operands", f32["nll"], nll_f32)
report("total: bf16 kernel nll - fp32 kernel nll", bf["nll"], f32["nll"])
dx = (bf["x_out"].double() - f32["x_out"].double())
print(f"x_out drift: rms {dx.pow(2).mean().sqrt().item():.3e} (x_out rms {f32['x_out'].double().pow(2).mean().sqrt().item():.3f}); mean {dx.mean().item():+.2e}")
# is the drift aligned with the target embedding direction?  d nll / d x = sum_j p_j w_j - w_tgt
Wt = f32["W"][tgt].double()
print(f"mean of drift . (-w_tgt) per token: {(-(dx * Wt).sum(1)).mean().item():+.3e}")

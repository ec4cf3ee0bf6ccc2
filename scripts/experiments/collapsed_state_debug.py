#!/usr/bin/env python3
"""Where does the default bf16 engine leave the fp32 engine at a hyper-collapsed state (fp32 engine trained 40 steps at B = 16)?
Per layer: common-mode (mean over the caption rows) and row-specific distance of the reconstructed h / sa from the fp32 engine's."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dic = importlib.import_module("diffusion-image-captioning_amd")
synth = dic.synth
B, S, L, V, nl = int(os.environ.get("BATCH", "16")), 1, 16, 30522, 12
dic.cfg.update(BATCH_SIZE=B, SAMPLE_SIZE=S, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, ROUNDING_WEIGHT=0.5, VOCAB_SIZE=V,
               LOSS_FUNC="series_sum_sample_mean", CLIP_ADDING_METHOD="concat", CLASSIFIER_FREE_WEIGHT=0.0, X_0_PREDICTION=True)
E = synth.vocab_embedding(V, 768, 0)
kw = dict(config=dict(n_layers=nl, dropout=0.0, attention_dropout=0.0))
m32 = dic.DistilBertModel(E, E, dtype="fp32", **kw)
m32.load_state(synth.denoiser_state(nl, 0))
tr = dic.AdamW(m32.parameters(), lr=1e-4)
batches = [synth.batch(B, L, V, 300 + i) for i in range(4)]
dic.seed_noise(99); dic.diffusion.seed_timesteps(77)
for i in range(int(os.environ.get("STEPS", "40"))):
    dic.train_func(m32, tr, {k: torch.from_numpy(v).cuda() for k, v in batches[i % 4].items()})
state = m32.state_dict()
held = {k: torch.from_numpy(v).cuda() for k, v in synth.batch(B, L, V, 9).items()}
t = torch.from_numpy(synth.timesteps(S, 100, 5))
nz = [torch.from_numpy(synth.noise((B, L, 768), 21, f"eps{i}")) for i in range(2)]
def run(m):
    m.eval()
    with torch.no_grad():
        out = [float(v) for v in dic.train_func(m, None, held, train=False, t=t, noises=nz)]
    ws = m._saved
    T, Tk = ws["T"], ws["Tk"]
    acts = {}
    refs = ws.get("refs") if ws.get("cen_fwd") else None
    for i in range(nl):
        h = ws["h"][i + 1][:T].float()
        sa = ws["layers"][i]["sa"][:T].float()
        if refs is not None:
            h = h + refs[i, 3]
            sa = sa + refs[i, 1]
        acts[f"L{i}.sa"], acts[f"L{i}.h"] = sa.double().cpu(), h.double().cpu()
    acts["x_out"] = ws["x_out"][:ws["N"]].reshape(-1, 768).double().cpu()
    return out, acts, Tk
ref_l, ref_a, Tk = run(m32)
cap = torch.tensor([r % Tk < L for r in range(ref_a["x_out"].shape[0])])
print("fp32 losses", ref_l)
for dt in os.environ.get("MODES", "bf16,bf16w,bf16r").split(","):
    m = dic.DistilBertModel(E, E, dtype=dt, **kw); m.load_state_dict(state)
    l, a, _ = run(m)
    print(f"== {dt}: loss rel", [f"{abs(x - y) / abs(y):.2e}" for x, y in zip(l, ref_l)])
    for k in list(a)[::4] + ["x_out"]:
        d = (a[k] - ref_a[k])[cap]
        r = ref_a[k][cap]
        cm = d.mean(0)
        print(f"   {k:8s} |ref row| {float(r.norm(dim=1).mean()):8.3f}  common-mode err {float(cm.norm()):.3e}  row-specific err (rms) {float((d - cm).norm(dim=1).pow(2).mean().sqrt()):.3e}"
              f"   spread of ref rows {float((r - r.mean(0)).norm(dim=1).pow(2).mean().sqrt()):.3e}")
    del m; torch.cuda.empty_cache()

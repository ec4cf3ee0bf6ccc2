#!/usr/bin/env python3
"""Is the bf16 engine's loss gap to the fp32 engine (round-3 review weak #1: 2.5e-4) made by the bf16 WEIGHTS or by the bf16 activations?

The host simulation (bf16_drift_sim.py) says: the weights.  Their rounding error is one fixed perturbation shared by every sample, so its
first-order effect on a batch-mean loss does not average out, while activation roundings are independent per element and do.  This probe
checks that on the real engines at the bench shape: the eval step (same batch / noise / timesteps, dropout off) in both dtypes
  (1) with the fp32 master weights as they are             -> the number bench.py prints as bf16_vs_fp32_loss_rel
  (2) with every weight rounded to bf16 beforehand          -> both engines see the same weights: what is left is the activations' share
for a few noise seeds.  DIC_SPLIT_W=1 (when the library has the two-pass forward) adds  (3) bf16 engine with hi+lo weights vs fp32 masters.
"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dic = importlib.import_module("diffusion-image-captioning_amd")
B, S, L, NL = int(os.environ.get("B", "512")), 1, 16, int(os.environ.get("LAYERS", "12"))
dic.cfg.update(BATCH_SIZE=B, SAMPLE_SIZE=S, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, VOCAB_SIZE=30522, CLASSIFIER_FREE_WEIGHT=0.0,
               CLIP_ADDING_METHOD="concat", LOSS_FUNC="series_sum_sample_mean", X_0_PREDICTION=True, ROUNDING_WEIGHT=0.5)
dev = "cuda:0"
E = torch.from_numpy(dic.synth.vocab_embedding(30522, 768, 0))
x = {k: torch.from_numpy(v).to(dev) for k, v in dic.synth.batch(B, L, 30522, seed=1).items()}
t = torch.from_numpy(dic.synth.timesteps(S, 100, 0))
kw = dict(config=dict(n_layers=NL, dropout=0.1, attention_dropout=0.1), device=dev, seed=0)
models = {"fp32": dic.DistilBertModel(E, E, dtype="fp32", **kw), "bf16": dic.DistilBertModel(E, E, dtype="bf16", **kw)}
have_split = "split_weights" in dic.DistilBertModel.__init__.__code__.co_varnames
if have_split:
    models["bf16+lo"] = dic.DistilBertModel(E, E, dtype="bf16", split_weights=True, **kw)
master = models["bf16"].state_dict()
rounded = {k: v.to(torch.bfloat16).to(torch.float32) for k, v in master.items()}


def losses(m, seed):
    nz = [torch.from_numpy(dic.synth.noise((B, L, 768), seed, f"eps{i}")) for i in range(2)]
    m.eval()
    with torch.no_grad():
        r = dic.train_func(m, None, x, train=False, t=t, noises=nz)
    return [float(v) for v in r]


def rel(a, b):
    return "  ".join(f"{n} {abs(p - q) / abs(q):.2e}" for n, p, q in zip(("total", "x_t", "x_1", "prob"), a, b))


for name, state in (("fp32 master weights", master), ("weights rounded to bf16 first", rounded)):
    for m in models.values():
        m.load_state_dict(state)
    for seed in (3, 4, 5):
        ref = losses(models["fp32"], seed)
        for k in models:
            if k != "fp32":
                print(f"{name:32s} noise seed {seed}  {k:8s} vs fp32:  {rel(losses(models[k], seed), ref)}", flush=True)

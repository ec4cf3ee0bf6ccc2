#!/usr/bin/env python3
"""Where does the 2.5e-4 rounding-loss gap between the bf16(w) and fp32 engines AFTER training on one batch come from (bench.py evaluates it after
its timed steps, all on the same synthetic batch)?  Hypothesis: not an arithmetic error of either engine -- the weights were optimised THROUGH the
bf16 engine's (nearly step-invariant, for the t = 1 rows) activation roundings on exactly these captions, so the engine they were trained in sees a
slightly lower loss than any other arithmetic does.  Test: evaluate the trained weights (a) on the training batch, (b) on a fresh batch, signed;
and train in fp32 instead, where the sign must flip."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dic = importlib.import_module("diffusion-image-captioning_amd")
B, S, L, NL = 512, 1, 16, 12
dic.cfg.update(BATCH_SIZE=B, SAMPLE_SIZE=S, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, VOCAB_SIZE=30522, CLASSIFIER_FREE_WEIGHT=0.0,
               CLIP_ADDING_METHOD="concat", LOSS_FUNC="series_sum_sample_mean", X_0_PREDICTION=True, ROUNDING_WEIGHT=0.5)
dev = "cuda:0"
E = torch.from_numpy(dic.synth.vocab_embedding(30522, 768, 0))
xs = {s: {k: torch.from_numpy(v).to(dev) for k, v in dic.synth.batch(B, L, 30522, seed=s).items()} for s in (1, 7)}
t = torch.from_numpy(dic.synth.timesteps(S, 100, 0))
nz = [torch.from_numpy(dic.synth.noise((B, L, 768), 3, f"eps{i}")) for i in range(2)]
kw = dict(config=dict(n_layers=NL, dropout=0.1, attention_dropout=0.1), device=dev, seed=0)


def ev(m, x):
    m.eval()
    with torch.no_grad():
        r = [float(v) for v in dic.train_func(m, None, x, train=False, t=t, noises=nz)]
    m.train()
    return r


models = {d: dic.DistilBertModel(E, E, dtype=d, **kw) for d in ("fp32", "bf16w", "bf16")}
for trainer_dtype, nstep in (("bf16w", 200), ("fp32", 60)):
    init = dic.DistilBertModel(E, E, dtype="fp32", **kw).state_dict()
    m = models[trainer_dtype]
    m.load_state_dict(init)
    tr = dic.AdamW(m.parameters(), lr=1e-4)
    for _ in range(nstep):
        dic.train_func(m, tr, xs[1])
    st = m.state_dict()
    for k in models:
        models[k].load_state_dict(st)
    for bname, seed in (("the training batch", 1), ("a fresh batch", 7)):
        ref = ev(models["fp32"], xs[seed])
        for k in ("bf16w", "bf16"):
            got = ev(models[k], xs[seed])
            print(f"trained {nstep:3d} steps in {trainer_dtype:5s}, evaluated on {bname:18s}: {k:5s} - fp32, relative: " +
                  "  ".join(f"{nm} {(a - b) / abs(b):+.2e}" for nm, a, b in zip(("total", "x_t", "x_1", "prob"), got, ref)), flush=True)

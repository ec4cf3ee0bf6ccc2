#!/usr/bin/env python3
"""Per-launch GEMM times INSIDE the training step (single stream, every kernel alone), grouped by position in the step: the same hipEvent
brackets bench.py's roofline leg uses (dic_prof_begin / dic_prof_get), so the numbers are those of hot-in-step operands, not of a cold
microbenchmark.  Prints one line per distinct launch of a step: shape signature (GFLOP, algorithmic MB), mean microseconds over the steps,
TFLOP/s and fraction of the 2.5 PFLOP/s bf16 peak.   python scripts/gemm_in_step.py [--dtype bf16|bf16w] [--steps 10]"""
import argparse, ctypes as C, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dic = importlib.import_module("diffusion-image-captioning_amd")
ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--layers", type=int, default=12)
ap.add_argument("--positions", action="store_true", help="also print every launch of the step in order")
a = ap.parse_args()
B, S, L = 512, 1, 16
dic.cfg.update(BATCH_SIZE=B, SAMPLE_SIZE=S, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, VOCAB_SIZE=30522, CLASSIFIER_FREE_WEIGHT=0.0,
               CLIP_ADDING_METHOD="concat", LOSS_FUNC="series_sum_sample_mean", X_0_PREDICTION=True, ROUNDING_WEIGHT=0.5)
dev = torch.device("cuda", 0)
E = dic.synth.vocab_embedding(30522, 768, 0)
model = dic.DistilBertModel(E, E, config=dict(n_layers=a.layers, dropout=0.1, attention_dropout=0.1), dtype=a.dtype, device=dev, seed=0)
trainer = dic.AdamW(model.parameters(), lr=1e-4)
x = {k: torch.from_numpy(v).to(dev) for k, v in dic.synth.batch(B, L, 30522, seed=1).items()}
for _ in range(3):
    dic.train_func(model, trainer, x)
Lh = dic.lib()
model.wgrad_stream_enabled = False
Lh.dic_prof_begin(a.steps * (a.layers * 16 + 64))
for _ in range(a.steps):
    dic.train_func(model, trainer, x)
torch.cuda.synchronize()
recs = []
ms, fl, by = C.c_double(), C.c_double(), C.c_double()
i = 0
while Lh.dic_prof_get(i, C.byref(ms), C.byref(fl), C.byref(by)):
    recs.append((ms.value, fl.value, by.value))
    i += 1
t, f, n = C.c_double(), C.c_double(), C.c_int()
Lh.dic_prof_end(C.byref(t), C.byref(f), C.byref(n))
per = len(recs) // a.steps
tot = 0.0
print(f"# {a.dtype}, {per} GEMM launches per step, mean over {a.steps} steps; position = order of the launch inside the step")
agg = {}
for pos in range(per):
    rs = [recs[s * per + pos] for s in range(a.steps)]
    us = sum(r[0] for r in rs) / a.steps * 1e3
    key = (round(rs[0][1] / 1e9, 2), round(rs[0][2] / 1e6, 1))
    agg.setdefault(key, []).append(us)
    tot += us
if a.positions:
    for pos in range(per):
        rs = [recs[s * per + pos] for s in range(a.steps)]
        print(f"pos {pos:3d}  {rs[0][1] / 1e9:8.2f} GFLOP {rs[0][2] / 1e6:8.1f} MB  {sum(r[0] for r in rs) / a.steps * 1e3:8.1f} us")
for (gf, mb), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    m = sum(v) / len(v)
    print(f"{gf:9.2f} GFLOP {mb:8.1f} MB  x{len(v):3d}/step  {m:8.1f} us  {gf / m * 1e3:7.1f} TFLOP/s  frac {gf / m * 1e3 / 2500:.3f}   {sum(v) / 1e3:6.3f} ms/step  ({mb / m:5.2f} TB/s algorithmic)")
print(f"total {tot / 1e3:.3f} ms/step")

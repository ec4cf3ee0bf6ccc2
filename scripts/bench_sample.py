#!/usr/bin/env python3
"""BASELINE.json config 4: the x0-prediction sampling loop (ref CLIP-DDPM.py:611-621) -- batch 2048 images, 100 encoder
passes, logits/argmax only after the last one.  Prints captions/s and ms per denoising pass."""
import argparse
import importlib
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dic = importlib.import_module("diffusion-image-captioning_amd")

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=2048)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--layers", type=int, default=12)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--bleu-batch", type=int, default=256)
a = ap.parse_args()
dic.cfg.update(MAX_LENGTH=16, CLASSIFIER_FREE_WEIGHT=0.0, CLIP_ADDING_METHOD="concat", VOCAB_SIZE=30522)
E = dic.synth.vocab_embedding(30522, 768, 0)
model = dic.DistilBertModel(E, E, config=dict(n_layers=a.layers), dtype=a.dtype)
model.eval()
img = torch.from_numpy(dic.synth.batch(a.batch, 16, 30522, 2)["image_clip"]).cuda()
dic.sample(model, img, steps=2)
torch.cuda.synchronize()
best = 1e9
for _ in range(a.reps):
    t0 = time.perf_counter()
    ids = dic.sample(model, img, steps=a.steps)
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
# BLEU-4 of the bf16 loop's ids against the fp32 loop's ids from the SAME start noise (the fp32 path is the one the -m gpu tests
# pin bit-exactly to the reference's ids on the golden fixture): how far 100 bf16 passes drift in token space
bleu = None
if a.dtype == "bf16" and a.bleu_batch > 0:
    nb = a.bleu_batch
    m32 = dic.DistilBertModel(E, E, config=dict(n_layers=a.layers), dtype="fp32")
    m32.load_state_dict(model.state_dict())
    m32.eval()
    start = torch.randn(nb, 18, 768, generator=torch.Generator().manual_seed(11)).cuda()
    ids16 = dic.sample(model, img[:nb], steps=a.steps, start=start).cpu()
    ids32 = dic.sample(m32, img[:nb], steps=a.steps, start=start).cpu()
    bleu = {"bleu4_bf16_vs_fp32_ids": round(dic.bleu.corpus_bleu([r.tolist() for r in ids16], [[r.tolist()] for r in ids32]), 4),
            "token_agreement": round(float((ids16 == ids32).float().mean()), 4), "captions": nb}
print(json.dumps({"metric": "sampling captions/sec", "value": round(a.batch / best, 1), "batch": a.batch, "denoising_steps": a.steps,
                  "n_layers": a.layers, "dtype": a.dtype, "ms_per_pass": round(best / a.steps * 1e3, 3), "loop_s": round(best, 3), "bleu": bleu}))

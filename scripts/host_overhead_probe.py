#!/usr/bin/env python3
"""How long does the HOST spend issuing one training step (ctypes launches + torch plumbing) vs the GPU executing it?"""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dic = importlib.import_module("diffusion-image-captioning_amd")
B, L = 512, 16
dic.cfg.update(BATCH_SIZE=B, SAMPLE_SIZE=1, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, VOCAB_SIZE=30522)
E = dic.synth.vocab_embedding(30522, 768, 0)
model = dic.DistilBertModel(E, E, config=dict(n_layers=12, dropout=0.1, attention_dropout=0.1), dtype="bf16")
trainer = dic.AdamW(model.parameters(), lr=1e-4)
x = {k: torch.from_numpy(v).cuda() for k, v in dic.synth.batch(B, L, 30522, 1).items()}
for _ in range(5): dic.train_func(model, trainer, x)
torch.cuda.synchronize()
# host issue time: launch 10 steps back-to-back, measure when the host is done vs when the GPU is done
t0 = time.perf_counter()
for _ in range(10): dic.train_func(model, trainer, x)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host issue {t_host/10*1e3:.2f} ms/step, GPU-complete {t_all/10*1e3:.2f} ms/step  -> host is {'NOT ' if t_host < 0.8*t_all else ''}the limiter")
# one step issued with an idle GPU: pure host time
torch.cuda.synchronize(); t0 = time.perf_counter(); dic.train_func(model, trainer, x); t1 = time.perf_counter() - t0; torch.cuda.synchronize()
print(f"single step host time with empty queue: {t1*1e3:.2f} ms")

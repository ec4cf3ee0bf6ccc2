#!/usr/bin/env python3
"""Soak run of the bf16 training step at the bench configuration (dropout 0.1, both streams, streamed AdamW): N steps over a few synthetic
batches; prints the loss terms every 100 steps, checks that nothing becomes non-finite and reports the steady-state step time.
    python scripts/soak.py [--steps 2000]"""
import argparse, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--batches", type=int, default=16)
    ap.add_argument("--dtype", default="bf16", help="bf16 | bf16m | bf16w")
    args = ap.parse_args()
    import torch
    dic = importlib.import_module("diffusion-image-captioning_amd")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    B, L = 512, 16
    dic.cfg.update(BATCH_SIZE=B, SAMPLE_SIZE=1, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, VOCAB_SIZE=30522, CLASSIFIER_FREE_WEIGHT=0.0,
                   CLASSIFIER_FREE_PROB=0.2, CLIP_ADDING_METHOD="concat", LOSS_FUNC="series_sum_sample_mean", X_0_PREDICTION=True, ROUNDING_WEIGHT=0.5)
    E = dic.synth.vocab_embedding(30522, 768, 0)
    data = [{k: torch.from_numpy(v).to(dev) for k, v in dic.synth.batch(B, L, 30522, seed=200 + i).items()} for i in range(args.batches)]
    model = dic.DistilBertModel(E, E, config=dict(n_layers=12, dropout=0.1, attention_dropout=0.1), dtype=args.dtype, device=dev, seed=0)
    trainer = dic.AdamW(model.parameters(), lr=1e-4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = []
    for s in range(args.steps):
        r = dic.train_func(model, trainer, data[s % args.batches])
        if s % 100 == 0 or s == args.steps - 1:
            outs.append((s, [float(v) for v in r]))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bad = 0
    for s, v in outs:
        ok = all(x == x and abs(x) < 1e30 for x in v)
        bad += not ok
        print(f"step {s:5d}: total {v[0]:10.4f}  x_t {v[1]:8.4f}  x_1 {v[2]:8.4f}  prob {v[3]:10.4f}{'' if ok else '   <-- non-finite'}")
    pn = float(torch.linalg.vector_norm(model.params.P.float()))
    print(f"{args.dtype}: {args.steps} steps in {dt:.1f} s = {dt / args.steps * 1e3:.3f} ms/step incl. the logging syncs; parameter norm {pn:.3f}; non-finite samples: {bad}")
    sys.exit(1 if bad or pn != pn else 0)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Training-loss curves of the bf16 (throughput) and fp32 (parity) engines from the same initial weights on the same batches, noise
streams and timestep draws: how far the benchmarked dtype drifts from the parity dtype over a run, not just at step 0 (bench.py's
`bf16_vs_fp32_loss_rel`).  Dropout is off (the two engines draw their keep-masks from different hash widths), everything else is the
bench configuration: B captions x S=1 (+ the x_1 pass), seq 16, n-layer denoiser, linear T=100, AdamW 1e-4.
    python scripts/loss_curve.py [--steps 200] [--batch 512] [--layers 12] [--batches 8]
"""
import argparse, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--batches", type=int, default=8, help="distinct synthetic batches cycled through (the model can fit them: the loss must fall)")
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--dtype", default="bf16", help="the engine compared with fp32: bf16 or bf16w")
    args = ap.parse_args()
    import torch
    dic = importlib.import_module("diffusion-image-captioning_amd")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    B, L = args.batch, 16
    dic.cfg.update(BATCH_SIZE=B, SAMPLE_SIZE=1, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, VOCAB_SIZE=30522, CLASSIFIER_FREE_WEIGHT=0.0,
                   CLASSIFIER_FREE_PROB=0.2, CLIP_ADDING_METHOD="concat", LOSS_FUNC="series_sum_sample_mean", X_0_PREDICTION=True, ROUNDING_WEIGHT=0.5)
    E = dic.synth.vocab_embedding(30522, 768, 0)
    data = [{k: torch.from_numpy(v).to(dev) for k, v in dic.synth.batch(B, L, 30522, seed=100 + i).items()} for i in range(args.batches)]
    curves = {}
    for dt in ("fp32", args.dtype):
        model = dic.DistilBertModel(E, E, config=dict(n_layers=args.layers, dropout=0.0, attention_dropout=0.0), dtype=dt, device=dev, seed=0)
        trainer = dic.AdamW(model.parameters(), lr=args.lr)
        dic.seed_noise(1234)                      # same eps stream for both engines
        dic.diffusion.seed_timesteps(4321)     # and the same timestep draws
        out = []
        for s in range(args.steps):
            r = dic.train_func(model, trainer, data[s % args.batches])
            out.append([float(v) for v in r])
        curves[dt] = out
        del model, trainer
        torch.cuda.empty_cache()
    names = ("total", "x_t", "x_1", "prob")
    print(f"# B={B} x S=1, {args.layers} layers, seq 16, linear T=100, dropout off, AdamW lr {args.lr}, {args.batches} synthetic batches cycled, {args.steps} steps")
    print(f"# step   fp32: total x_t x_1 prob   |   {args.dtype}: total x_t x_1 prob   |   rel. diff of total")
    worst = 0.0
    for s in range(args.steps):
        a, b = curves["fp32"][s], curves[args.dtype][s]
        rel = abs(b[0] - a[0]) / abs(a[0])
        worst = max(worst, rel)
        if s < 10 or s % 10 == 0 or s == args.steps - 1:
            print(f"{s:5d}   " + " ".join(f"{v:10.4f}" for v in a) + "   |   " + " ".join(f"{v:10.4f}" for v in b) + f"   |   {rel:.2e}")
    f0, f1, b1 = curves["fp32"][0][0], curves["fp32"][-1][0], curves[args.dtype][-1][0]
    print(f"# total loss {f0:.4f} -> fp32 {f1:.4f}, {args.dtype} {b1:.4f} after {args.steps} steps; max relative difference of the total over the run {worst:.2e}")
    for k, nm in enumerate(names):
        print(f"#   {nm:5s}: fp32 {curves['fp32'][0][k]:.4f} -> {curves['fp32'][-1][k]:.4f}   {args.dtype} {curves[args.dtype][0][k]:.4f} -> {curves[args.dtype][-1][k]:.4f}")


if __name__ == "__main__":
    main()

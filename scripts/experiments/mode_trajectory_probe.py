#!/usr/bin/env python3
"""Loss distance from the fp32 engine of every bf16 mode ALONG one training run at the bench shape (the run of split_alloc_probe.py: the split-weight
engine trains on 8 cycled batches; evaluation on two held-out batches x three noise / timestep draws, dropout off):
  bf16r  raw (fp32 MLM-head pre-activation, centred head, nothing else)          bf16-r4  round 4's parity mode ("bf16m" then): mean-row lo correction (dic_lo_mean_bias) + fp32 residual stream
  bf16   the default = the parity mode: dic_lin_prep + CENTRED bf16 residual stream  bf16w  second K-loop pass against the lo halves + fp32 residual stream
    python scripts/experiments/mode_trajectory_probe.py [--trajectory 0,1,2,...] [--time]
"""
import argparse, importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dic = importlib.import_module("diffusion-image-captioning_amd")
eng = importlib.import_module("diffusion-image-captioning_amd.engine")
ap = argparse.ArgumentParser()
ap.add_argument("--trajectory", default="0,1,2,3,5,7,10,15,20,30,40,60,80,120,160,240,320,480,640")
ap.add_argument("--layers", type=int, default=12)
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--time", action="store_true", help="also time 20 training steps of each mode (dropout 0.1) at the end")
ap.add_argument("--train-dropout", type=float, default=0.0, help="dropout of the TRAINING steps (the evaluations always run with dropout off)")
ap.add_argument("--lr", type=float, default=1e-4)
ap.add_argument("--seed-offset", type=int, default=0, help="shifts every data / noise / timestep seed: an independent run")
args = ap.parse_args()
B, S, L, NL = args.batch, 1, 16, args.layers
dic.cfg.update(BATCH_SIZE=B, SAMPLE_SIZE=S, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, VOCAB_SIZE=30522, CLASSIFIER_FREE_WEIGHT=0.0,
               CLIP_ADDING_METHOD="concat", LOSS_FUNC="series_sum_sample_mean", X_0_PREDICTION=True, ROUNDING_WEIGHT=0.5)
dev = "cuda:0"
E = torch.from_numpy(dic.synth.vocab_embedding(30522, 768, 0))


def make(dtype, res32=None, drop=0.0, cen=None):
    keep = eng.OPT.res32, eng.OPT.cen
    if res32 is not None:
        eng.OPT.res32 = res32
    if cen is not None:
        eng.OPT.cen = cen
    try:
        return dic.DistilBertModel(E, E, dtype=dtype, config=dict(n_layers=NL, dropout=drop, attention_dropout=drop), device=dev, seed=0)
    finally:
        eng.OPT.res32, eng.OPT.cen = keep


# bf16m: round 5's parity mode (centred bf16 residual stream, dic_lin_prep); bf16m-r4: round 4's (fp32 residual stream, dic_lo_mean_bias)
MODES = {"bf16r": make("bf16r"), "bf16-r4": make("bf16", cen=False), "bf16": make("bf16"), "bf16w": make("bf16w", drop=args.train_dropout)}
SO = args.seed_offset
f32 = make("fp32")
held = [{k: torch.from_numpy(v).to(dev) for k, v in dic.synth.batch(B, L, 30522, seed=1 + 7 * i + SO).items()} for i in range(2)]
train = [{k: torch.from_numpy(v).to(dev) for k, v in dic.synth.batch(B, L, 30522, seed=100 + i + SO).items()} for i in range(8)]
draws = [(torch.from_numpy(dic.synth.timesteps(S, 100, i + SO)), [torch.from_numpy(dic.synth.noise((B, L, 768), 3 + i + SO, f"eps{j}")) for j in range(2)]) for i in range(3)]


def evals(m):
    m.eval()
    out = []
    with torch.no_grad():
        for x in held:
            for t, nz in draws:
                out.append([float(v) for v in dic.train_func(m, None, x, train=False, t=t, noises=nz)])
    return torch.tensor(out, dtype=torch.float64)


bw = MODES["bf16w"]
trainer = dic.AdamW(bw.parameters(), lr=args.lr)
dic.seed_noise(1234 + SO)
dic.diffusion.seed_timesteps(4321 + SO)
print(f"# run: lr {args.lr}, training dropout {args.train_dropout}, seed offset {SO}")
done = 0
worst = {k: 0.0 for k in MODES}
inside = {k: 0 for k in MODES}
states = [int(v) for v in args.trajectory.split(",")]
for upto in states:
    bw.train()
    while done < upto:
        dic.train_func(bw, trainer, train[done % 8])
        done += 1
    state = bw.state_dict()
    f32.load_state_dict(state)
    ref = evals(f32)
    line = f"after {done:4d} steps (x_t {float(ref[0, 1]):7.4f}):"
    for name, m in MODES.items():
        if m is not bw:
            m.load_state_dict(state)
        rel = ((evals(m) - ref).abs() / ref.abs()).max(0).values
        w = float(rel.max())
        worst[name] = max(worst[name], w)
        inside[name] += w <= 1e-4
        line += f"   {name} x_t {float(rel[1]):.1e} x_1 {float(rel[2]):.1e} prob {float(rel[3]):.1e}{' ' if w <= 1e-4 else '*'}"
    print(line, flush=True)
print("# worst term over the run / states with every term <= 1e-4:  " + "   ".join(f"{k}: {worst[k]:.2e}, {inside[k]}/{len(states)}" for k in MODES))
if args.time:
    del f32
    x = train[0]
    for name in ("bf16r", "bf16-r4", "bf16", "bf16w", "bf16r", "bf16"):
        m = make({"bf16-r4": "bf16"}.get(name, name), cen=False if name == "bf16-r4" else None, drop=0.1)
        tr = dic.AdamW(m.parameters(), lr=1e-4)
        for _ in range(5):
            dic.train_func(m, tr, x)
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        for _ in range(30):
            dic.train_func(m, tr, x)
        torch.cuda.synchronize()
        print(f"# {name:8s} {(time.perf_counter() - c0) / 30 * 1e3:7.3f} ms per training step", flush=True)
        del m, tr
        torch.cuda.empty_cache()

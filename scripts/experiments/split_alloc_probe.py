#!/usr/bin/env python3
"""Which forward Linears need the lo halves of their weights for the bf16 engine to stay inside north_star's 1e-4 loss tolerance?

dtype="bf16w" adds the lo half to EVERY forward Linear (two K-loop passes, +2.2 ms of a 13.8 ms step).  The weights' rounding error enters a
batch-mean loss at first order as <dL/dW, W - bf16(W)>, so the share of a matrix is set by the gradient that reaches it.  This probe switches
the lo halves on per slot (Denoiser.split_slots) and prints, against the fp32 engine on the same weights / batch / noise / timesteps
(eval, dropout off), the worst relative distance of each loss term over a few noise seeds and batches:
  state A: the initial weights;  state B: after --train-steps AdamW steps over 8 cycled batches, evaluated on batches NOT trained on.
    python scripts/experiments/split_alloc_probe.py [--train-steps 300]
"""
import argparse, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dic = importlib.import_module("diffusion-image-captioning_amd")
ap = argparse.ArgumentParser()
ap.add_argument("--train-steps", type=int, default=300)
ap.add_argument("--layers", type=int, default=12)
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--seeds", type=int, default=3)
ap.add_argument("--only", default="", help="comma-separated substrings: keep only the SHORT configs whose name contains one of them")
ap.add_argument("--trajectory", default="", help="comma-separated step counts: the short config list at each of these training states instead")
args = ap.parse_args()
B, S, L, NL = args.batch, 1, 16, args.layers
dic.cfg.update(BATCH_SIZE=B, SAMPLE_SIZE=S, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, VOCAB_SIZE=30522, CLASSIFIER_FREE_WEIGHT=0.0,
               CLIP_ADDING_METHOD="concat", LOSS_FUNC="series_sum_sample_mean", X_0_PREDICTION=True, ROUNDING_WEIGHT=0.5)
dev = "cuda:0"
E = torch.from_numpy(dic.synth.vocab_embedding(30522, 768, 0))
kw = dict(config=dict(n_layers=NL, dropout=0.0, attention_dropout=0.0), device=dev, seed=0)
f32 = dic.DistilBertModel(E, E, dtype="fp32", **kw)
bw = dic.DistilBertModel(E, E, dtype="bf16w", **kw)
held_out = [{k: torch.from_numpy(v).to(dev) for k, v in dic.synth.batch(B, L, 30522, seed=1 + 7 * i).items()} for i in range(2)]
t_draws = [torch.from_numpy(dic.synth.timesteps(S, 100, i)) for i in range(args.seeds)]


def layer_of(slot):
    return int(slot[1:slot.index(".")]) if slot.startswith("L") else NL          # "Wvt" counts as the layer after the last


CONFIGS = [("none (= bf16 + centred head)", lambda s: False), ("all", lambda s: True), ("default set (= bf16w)", None)]
for k in (1, 2, 3, 4, 6, 8):
    CONFIGS.append((f"last {k} layers + Wvt", (lambda k: lambda s: layer_of(s) >= NL - k)(k)))
CONFIGS.append(("Wvt only", lambda s: s == "Wvt"))
for k in (2, 4, 6):
    CONFIGS.append((f"first {k} layers", (lambda k: lambda s: layer_of(s) < k)(k)))
CONFIGS += [("FFN (W1, W2) only", lambda s: s.endswith("W1") or s.endswith("W2")),
            ("attention (Wqkv, Wo) + Wvt only", lambda s: s.endswith("Wqkv") or s.endswith("Wo") or s == "Wvt"),
            ("residual writers (Wo, W2) + Wvt", lambda s: s.endswith("Wo") or s.endswith("W2") or s == "Wvt"),
            ("W1 only", lambda s: s.endswith("W1")), ("W2 only", lambda s: s.endswith("W2")),
            ("Wo only", lambda s: s.endswith("Wo")), ("Wqkv only", lambda s: s.endswith("Wqkv"))]


is_attn = lambda s: s.endswith("Wqkv") or s.endswith("Wo")
# (name, slot predicate, zero the lo halves of the q / k rows of Wqkv first: "v" = only the value projection of Wqkv keeps its lo half)
SHORT = [("none (= bf16 + centred head)", lambda s: False, False), ("all", lambda s: True, False), ("default set (= bf16w)", None, False),
         ("Wqkv + Wo + Wvt", lambda s: is_attn(s) or s == "Wvt", False), ("Wqkv + Wo", is_attn, False),
         ("Wv + Wo + Wvt", lambda s: is_attn(s) or s == "Wvt", True), ("Wv + Wo", is_attn, True),
         ("Wq,Wk + Wvt", lambda s: s.endswith("Wqkv") or s == "Wvt", "qk"),
         ("Wo + Wvt", lambda s: s.endswith("Wo") or s == "Wvt", False), ("Wqkv + Wvt", lambda s: s.endswith("Wqkv") or s == "Wvt", False),
         ("Wqkv + Wo + Wvt + W2", lambda s: is_attn(s) or s == "Wvt" or s.endswith("W2"), False),
         ("all but W1", lambda s: not s.endswith("W1"), False), ("all but W1, Wq, Wk", lambda s: not s.endswith("W1"), True),
         ("all but W1, W2", lambda s: not (s.endswith("W1") or s.endswith("W2")), False),
         ("last 4 layers + Wvt", lambda s: layer_of(s) >= NL - 4, False),
         ("Wqkv + Wo of last 6 + Wvt", lambda s: (is_attn(s) and layer_of(s) >= NL - 6) or s == "Wvt", False)]


if args.only:
    SHORT = [c for c in SHORT if any(k in c[0] for k in args.only.split(","))]


def mask_qkv_lo(mode):
    """mode True: keep only the value rows of every Wqkv's lo half; "qk": keep only the query / key rows; False: restore."""
    bw.refresh_shadows()
    if mode:
        for i in range(NL):
            v = bw.params.slot_view(bw.params.Pl, f"L{i}.Wqkv")
            if mode == "qk":
                v[2 * 768:].zero_()
            else:
                v[:2 * 768].zero_()


def losses(m, x, seed):
    nz = [torch.from_numpy(dic.synth.noise((B, L, 768), 3 + seed, f"eps{i}")) for i in range(2)]
    m.eval()
    with torch.no_grad():
        r = dic.train_func(m, None, x, train=False, t=t_draws[seed], noises=nz)
    return [float(v) for v in r]


def sweep(title):
    print(f"## {title}", flush=True)
    refs = {(bi, s): losses(f32, x, s) for bi, x in enumerate(held_out) for s in range(args.seeds)}
    print("   fp32 losses (batch 0, seed 0): " + " ".join(f"{v:.4f}" for v in refs[(0, 0)]))
    for name, sel, *rest in (SHORT if args.trajectory else CONFIGS):
        bw.split_slots = sel
        mask_qkv_lo(rest[0] if rest else False)
        worst = [0.0] * 4
        for (bi, s), ref in refs.items():
            got = losses(bw, held_out[bi], s)
            worst = [max(w, abs(p - q) / abs(q)) for w, p, q in zip(worst, got, ref)]
        flag = "ok " if max(worst) <= 1e-4 else "   "
        print(f"   {flag}{name:36s} worst of {len(refs)}: " + "  ".join(f"{n} {w:.2e}" for n, w in zip(("total", "x_t", "x_1", "prob"), worst)), flush=True)
    bw.split_slots = None
    mask_qkv_lo(False)


f32.load_state_dict(bw.state_dict())
sweep("state A: initial weights")
if args.trajectory:
    train = [{k: torch.from_numpy(v).to(dev) for k, v in dic.synth.batch(B, L, 30522, seed=100 + i).items()} for i in range(8)]
    trainer = dic.AdamW(bw.parameters(), lr=1e-4)
    dic.seed_noise(1234)
    dic.diffusion.seed_timesteps(4321)
    done = 0
    for upto in [int(v) for v in args.trajectory.split(",")]:
        bw.train()
        while done < upto:
            r = dic.train_func(bw, trainer, train[done % 8])
            done += 1
        print(f"# after {done} steps on 8 cycled batches: last training losses " + " ".join(f"{float(v):.4f}" for v in r))
        f32.load_state_dict(bw.state_dict())
        sweep(f"after {done} steps, held-out batches")
elif args.train_steps:
    train = [{k: torch.from_numpy(v).to(dev) for k, v in dic.synth.batch(B, L, 30522, seed=100 + i).items()} for i in range(8)]
    trainer = dic.AdamW(bw.parameters(), lr=1e-4)
    bw.train()
    dic.seed_noise(1234)
    dic.diffusion.seed_timesteps(4321)
    for s in range(args.train_steps):
        r = dic.train_func(bw, trainer, train[s % 8])
    print(f"# trained {args.train_steps} steps on 8 cycled batches: last losses " + " ".join(f"{float(v):.4f}" for v in r))
    f32.load_state_dict(bw.state_dict())
    sweep(f"state B: after {args.train_steps} steps, held-out batches")
    held_out[:] = train[:2]
    sweep(f"state B': after {args.train_steps} steps, two of the batches trained on")

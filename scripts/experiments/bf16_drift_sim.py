#!/usr/bin/env python3
"""Host-side what-if: which bf16 rounding points of the engine cost how much of north_star's 1e-4 loss tolerance.

Self-contained fp32 torch restatement of the eval step at the bench shape (same synthetic weights / batch / noise / timesteps as
bench.py's `bf16_vs_fp32_loss_rel` leg) with a switchable bf16 round-trip at every place where the HIP engine stores or feeds bf16:
weights (w), GEMM operands (h_op, sa_op, ctx, g, xr, wlm), attention internals (qkv, p), residual reads (h_res, sa_res), pre-LayerNorm sums
(y1, y2) and the MLM-head pre-activation (uvt).  GEMMs accumulate in fp32 either way, like the MFMA.  Prints the relative distance of the
three loss terms from the all-fp32 run for each named set of rounding points.

    python scripts/experiments/bf16_drift_sim.py [--batch 512] [--layers 12] [--variants cur,res32,...]

Round 4 used this to decide what to build before spending GPU time (results: profiles/r04_bf16_drift_sim.txt).
"""
import argparse
import importlib
import math
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
synth = importlib.import_module("diffusion-image-captioning_amd.synth")

ALL = {"w", "h_op", "h_res", "qkv", "p", "ctx", "y1", "sa_op", "sa_res", "g", "y2", "uvt", "xr", "wlm"}
VARIANTS = {
    "fp32": set(),
    "cur": set(ALL),                                                        # the round-3 bf16 engine
    "res32": ALL - {"h_res", "sa_res", "y1", "y2"},                         # fp32 residual stream, bf16 MFMA operands
    "res32_uvt": ALL - {"h_res", "sa_res", "y1", "y2", "uvt"},
    "res32_uvt_qkv": ALL - {"h_res", "sa_res", "y1", "y2", "uvt", "qkv"},
    "res32_now": ALL - {"h_res", "sa_res", "y1", "y2", "uvt", "w"},          # + split (hi+lo) weights
    "res32_noact": {"w", "wlm"},                                              # only the weights rounded
    "only_ops": {"h_op", "sa_op", "ctx", "g", "xr"},                          # only the activation operands rounded
    "only_head": {"xr", "wlm"},                                               # bf16 only in the rounding head
    "res32_head32": ALL - {"h_res", "sa_res", "y1", "y2", "uvt", "xr", "wlm"},
}


_SR_RUNS = 0


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def bf_sr(x, gen):
    """Stochastic rounding to bf16 (round 5): add 16 uniform random bits below the bf16 mantissa, truncate.  Unbiased, and -- unlike
    round-to-nearest -- independent from row to row even when the rows are equal, which is what a batch-mean loss averages out."""
    bits = x.contiguous().view(torch.int32)
    rnd = torch.randint(0, 65536, bits.shape, dtype=torch.int32, device=x.device, generator=gen)
    return ((bits + rnd) & -65536).view(torch.float32)


def run(points, P, E, batch, t, nz, layers, L=16):
    # "sr" (round 5): every ACTIVATION rounding point that is on rounds stochastically (one draw per stored tensor: h / sa feed the GEMM and
    # the residual add from the same bf16 copy); "p" (in-register attention probabilities) and the weights keep round-to-nearest
    sr = "sr" in points
    global _SR_RUNS
    _SR_RUNS += 1                                    # every stochastic run is a fresh draw (listing a group twice shows the spread)
    gen = torch.Generator(device=E.device).manual_seed(77 + _SR_RUNS) if sr else None
    memo = {}

    def r(name, x):
        if name not in points:
            return x
        if sr and name not in ("w", "wlm", "p"):
            k = id(x)
            if k not in memo:
                memo[k] = (x, bf_sr(x, gen))          # (keeps x alive so the id stays unique)
            return memo[k][1]
        return bf(x)
    W = lambda n: r("w", P[n])

    def lin(x, wname, bname, abar=None):
        """nn.Linear with the weight rounded to bf16 when "w" is on.  "wcm" / "wcmp" (round 4, collapse_probe.py): add back the part of the
        rounding's effect that is common to the rows -- mean row (of all rows / of the rows at each sequence position of each pass) times the
        lo half: one GEMV (or a [2T x K] GEMM) per Linear instead of a second pass over every row."""
        y = F.linear(x, W(wname), P[bname])
        if "w" in points and ("wcm" in points or "wcmp" in points):
            lo = P[wname] - bf(P[wname])
            if "wcmp" in points and x.dim() == 3:
                half = x.shape[0] // 2
                xm = torch.cat([x[:half].mean(0, keepdim=True).expand(half, -1, -1), x[half:].mean(0, keepdim=True).expand(x.shape[0] - half, -1, -1)])
                y = y + F.linear(xm, lo)
            else:             # abar (round 5, "wcmref"): a PREDICTED mean row (the centred stream's reference row) instead of the measured one
                y = y + F.linear(x.reshape(-1, x.shape[-1]).mean(0) if abar is None else abar, lo)
        return y
    B = batch["input_ids"].shape[0]
    dev = E.device                                                          # (round 4: the same what-if on the GPU, collapse_probe.py)
    betas = torch.hstack([torch.zeros(1), torch.linspace(1e-4, 0.02, 100)]).to(dev)
    ac = torch.cumprod((1 - betas)[:-1], 0)
    x0 = E[batch["input_ids"]]

    def diffuse(x, tt, eps):
        return torch.sqrt(ac[tt]).reshape(-1, 1, 1) * x + eps * torch.sqrt(1 - ac[tt]).reshape(-1, 1, 1)
    x_t = diffuse(x0, t.reshape(-1), nz[0])
    x_1 = diffuse(x0, torch.ones(1, dtype=torch.int64, device=dev), nz[1])
    x = torch.cat([x_t, x_1])                                               # one stacked batch, like the engine
    n = x.shape[0]
    img = F.linear(batch["image_clip"].repeat(2, 1), P["image_linear.weight"], P["image_linear.bias"])
    mask = torch.cat([batch["attention_mask"].repeat(2, 1), torch.ones(n, 1, dtype=torch.int64, device=dev)], 1)       # text row dropped (masked, never read)
    seg = P["segment_embedding.weight"]
    pre = "model.distilbert."
    h = torch.cat([x + seg[0], (img + seg[1]).unsqueeze(1)], 1) + P[pre + "embeddings.position_embeddings.weight"][:L + 1]
    ln = lambda v, a: F.layer_norm(v, (768,), P[a + ".weight"], P[a + ".bias"], 1e-12)
    h = ln(h, pre + "embeddings.LayerNorm")
    T = L + 1
    # "cen" (round 5): the residual stream stored as bf16(value - reference row), the reference row (fp32, one per tensor) PREDICTED before the
    # tensor exists: pre-LayerNorm sums y_ref = (mean input row) W^T + b + residual's reference, LayerNorm outputs o_ref = LN(y_ref), the
    # embedding LayerNorm's output uncentred (its rows are noised embeddings, never equal).  "cenop": the GEMM operand copy of h / sa is
    # that same centred tensor (its reference row enters through the bias: o_ref W^T in fp32).
    cen, cenop = "cen" in points, "cenop" in points
    cb = lambda v, ref: ref + bf(v - ref)
    h_ref = torch.zeros(768, device=dev)
    for i in range(layers):
        lp = pre + f"transformer.layer.{i}."
        if cen:
            h_st = cb(h, h_ref)
            ho = h_st if cenop else r("h_op", h)
        else:
            ho = r("h_op", h)
        pred = "wcmref" in points and cen                  # ("wcmref_qkv" / "_lin1" / "_vt": one family of LayerNorm-fed Linears at a time)
        pq = (pred or "wcmref_qkv" in points) and cen and i > 0
        q, k, v = (r("qkv", lin(ho, lp + f"attention.{s}_lin.weight", lp + f"attention.{s}_lin.bias", h_ref if pq else None)) for s in "qkv")
        sh = lambda z: z.view(n, T, 12, 64).transpose(1, 2)
        s_ = torch.matmul(sh(q), sh(k).transpose(2, 3)) * 0.125
        s_ = s_.masked_fill(mask[:, None, None, :] == 0, float("-inf"))
        p_ = r("p", F.softmax(s_, -1))
        ctx = r("ctx", torch.matmul(p_, sh(v)).transpose(1, 2).reshape(n, T, 768))
        if cen:
            flat = lambda z: z.reshape(-1, z.shape[-1])[::16].mean(0)                # the engine's sampled mean row (every 16th token row)
            y1_ref = F.linear(flat(ctx), P[lp + "attention.out_lin.weight"], P[lp + "attention.out_lin.bias"]) + h_ref
            y1 = cb(lin(ctx, lp + "attention.out_lin.weight", lp + "attention.out_lin.bias") + h_st, y1_ref)
            sa = ln(y1, lp + "sa_layer_norm")
            sa_ref = ln(y1_ref, lp + "sa_layer_norm")
            sa_st = cb(sa, sa_ref)
            g = r("g", F.gelu(lin(sa_st if cenop else r("sa_op", sa), lp + "ffn.lin1.weight", lp + "ffn.lin1.bias", sa_ref if (pred or "wcmref_lin1" in points) else None)))
            y2_ref = F.linear(flat(g), P[lp + "ffn.lin2.weight"], P[lp + "ffn.lin2.bias"]) + sa_ref
            y2 = cb(lin(g, lp + "ffn.lin2.weight", lp + "ffn.lin2.bias") + sa_st, y2_ref)
            h = ln(y2, lp + "output_layer_norm")
            h_ref = ln(y2_ref, lp + "output_layer_norm")
            continue
        y1 = r("y1", lin(ctx, lp + "attention.out_lin.weight", lp + "attention.out_lin.bias") + r("h_res", h))
        sa = ln(y1, lp + "sa_layer_norm")
        g = r("g", F.gelu(lin(r("sa_op", sa), lp + "ffn.lin1.weight", lp + "ffn.lin1.bias")))
        y2 = r("y2", lin(g, lp + "ffn.lin2.weight", lp + "ffn.lin2.bias") + r("sa_res", sa))
        h = ln(y2, lp + "output_layer_norm")
    if cen and cenop:
        h = cb(h, h_ref)
    u = r("uvt", lin(h if (cen and cenop) else r("h_op", h), "model.vocab_transform.weight", "model.vocab_transform.bias", h_ref if (cen and ("wcmref" in points or "wcmref_vt" in points)) else None))
    xo = ln(F.gelu(u), "model.vocab_layer_norm")[:, :L]
    tgt = x0.repeat(2, 1, 1)
    l1 = (xo - tgt).abs().sum(1)
    xt_loss, x1_loss = l1[:B].mean(), l1[B:].mean()
    Wl = r("wlm", E)
    ids = batch["input_ids"].repeat(2, 1)
    nll = torch.empty(n, L, device=dev)
    xr = r("xr", xo)
    for a in range(0, n, 128):                                             # logits in row blocks (never 2 GB at once)
        lg = xr[a:a + 128].reshape(-1, 768) @ Wl.t()
        nll[a:a + 128] = -(F.log_softmax(lg, -1).gather(-1, ids[a:a + 128].reshape(-1, 1))).reshape(-1, L)
    prob = 0.5 * (nll[:B].sum(1).mean() + nll[B:].sum(1).mean())
    return [float(v) for v in (xt_loss + x1_loss + prob, xt_loss, x1_loss, prob)], xo


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--seed", type=int, default=3, help="noise seed (bench.py uses 3)")
    ap.add_argument("--variants", default="cur,res32,res32_uvt")
    a = ap.parse_args()
    torch.manual_seed(0)
    P = {k: torch.from_numpy(v) for k, v in synth.denoiser_state(a.layers, 0).items()}
    E = torch.from_numpy(synth.vocab_embedding(30522, 768, 0))
    batch = {k: torch.from_numpy(v) for k, v in synth.batch(a.batch, 16, 30522, 1).items()}
    t = torch.from_numpy(synth.timesteps(1, 100, 0))
    nz = [torch.from_numpy(synth.noise((a.batch, 16, 768), a.seed, f"eps{i}")) for i in range(2)]
    with torch.no_grad():
        c0 = time.time()
        ref, xo_ref = run(set(), P, E, batch, t, nz, a.layers)
        print(f"# B={a.batch} layers={a.layers} noise seed {a.seed}; fp32 losses total/x_t/x_1/prob = {ref}  ({time.time() - c0:.0f} s)", flush=True)
        for name in a.variants.split(","):
            pts = VARIANTS[name] if name in VARIANTS else set(name.split("+"))
            v, xo = run(pts, P, E, batch, t, nz, a.layers)
            rel = [abs(x - y) / abs(y) for x, y in zip(v, ref)]
            drift = float(((xo - xo_ref) ** 2).mean().sqrt() / (xo_ref ** 2).mean().sqrt())
            print(f"{name:16s} rel total {rel[0]:.2e}  x_t {rel[1]:.2e}  x_1 {rel[2]:.2e}  prob {rel[3]:.2e}   x_out drift rms {drift:.2e}", flush=True)


if __name__ == "__main__":
    main()

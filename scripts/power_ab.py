#!/usr/bin/env python3
"""Round-4 review item 2: prove or kill "the step runs at the package power limit, so a faster GEMM is paid back in clock".

One process per arm (the library's switches are process-global); each arm loops the training step for a few seconds and reports
  (c) ms per step, mean socket power, rocm-smi shader clock, JOULES per step = mean power x step time;
  (a) with the clock-probe build (abl/libdic_clk.so, scripts/build_variant.sh clk "-DDIC_CLOCK_PROBE"): the shader clock realised INSIDE every GEMM launch
      of the step (s_memtime / s_memrealtime stamped by workgroup 0 at its start and end), averaged per kernel shape, next to that launch's duration.
Arms: the 8-wave kernel everywhere (the default of rounds 1-4) / the four-wave asm kernel for every eligible launch (options.gemm_w4a), each in the raw engine (what round 4
measured) -- and, when --pin is given, again under a pinned clock (b): `rocm-smi --setperfdeterminism MHZ` (falls back to --setsclk / a lowered power cap;
prints which one the box accepted, or that it refused all of them).

    python scripts/power_ab.py [--steps 400] [--pin 1900] [--dtype bf16r]          (driver: runs every arm as a subprocess of itself)
"""
import argparse
import importlib
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def smi(args):
    try:
        return subprocess.run(["rocm-smi"] + args, capture_output=True, text=True, timeout=20).stdout
    except Exception as e:
        return f"rocm-smi failed: {e}"


def arm(args):
    import torch
    import bench
    dic = importlib.import_module("diffusion-image-captioning_amd")
    L = dic.lib()
    dev = torch.device("cuda", 0)
    B, Ln = 512, 16
    dic.cfg.update(BATCH_SIZE=B, SAMPLE_SIZE=1, MAX_LENGTH=Ln, STEP_TOT=100, COSIN_SCHEDULE=False, VOCAB_SIZE=30522, CLASSIFIER_FREE_WEIGHT=0.0,
                   CLIP_ADDING_METHOD="concat", LOSS_FUNC="series_sum_sample_mean", X_0_PREDICTION=True, ROUNDING_WEIGHT=0.5)
    E = dic.synth.vocab_embedding(30522, 768, 0)
    model = dic.DistilBertModel(E, E, config=dict(n_layers=12, dropout=0.1, attention_dropout=0.1), dtype=args.dtype, device=dev, seed=0)
    trainer = dic.AdamW(model.parameters(), lr=1e-4)
    x = {k: torch.from_numpy(v).to(dev) for k, v in dic.synth.batch(B, Ln, 30522, seed=1).items()}
    for _ in range(20):
        dic.train_func(model, trainer, x)
    torch.cuda.synchronize()
    with bench.PowerSampler(0, period=0.4) as ps:
        c0 = time.perf_counter()
        for _ in range(args.steps):
            dic.train_func(model, trainer, x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - c0) / args.steps
    pw = ps.summary() or {}
    out = {"ms_per_step": round(dt * 1e3, 3), "power_w": pw.get("socket_power_w_mean"), "power_w_max": pw.get("socket_power_w_max"), "cap_w": pw.get("power_cap_w"),
           "sclk_mhz": pw.get("sclk_mhz_mean"), "joules_per_step": round(pw["socket_power_w_mean"] * dt, 2) if pw.get("socket_power_w_mean") else None}
    if hasattr(L, "dic_clock_probe_set"):
        import ctypes
        L.dic_clock_probe_set.argtypes = [ctypes.c_void_p, ctypes.c_int]
        # per-launch clock: a few more steps with the probe armed (every GEMM launch writes one 64-byte record)
        nsteps, cap = 12, 12 * 400
        buf = torch.zeros(cap * 8, dtype=torch.int64, device=dev)
        assert L.dic_clock_probe_set(buf.data_ptr(), cap) == 0
        for _ in range(nsteps):
            dic.train_func(model, trainer, x)
        torch.cuda.synchronize()
        n = min(L.dic_clock_probe_count(), cap)
        L.dic_clock_probe_set(0, 0)
        rec = buf.cpu().view(cap, 8)[:n].numpy()
        groups = {}
        for r in rec:
            if r[2] == 0 or r[3] <= r[1]:
                continue
            key = (int(r[7]), int(r[4]), int(r[5]), int(r[6]))
            us = (r[3] - r[1]) / 100.0                            # s_memrealtime ticks at 100 MHz
            mhz = (r[2] - r[0]) / us
            g = groups.setdefault(key, [0, 0.0, 0.0])
            g[0] += 1; g[1] += us; g[2] += mhz
        out["launches_probed"] = int(n)
        out["kernels"] = [{"tag": k[0], "M": k[1], "N": k[2], "K": k[3], "per_step": round(v[0] / nsteps, 1), "us": round(v[1] / v[0], 1), "mhz_in_kernel": round(v[2] / v[0])}
                          for k, v in sorted(groups.items(), key=lambda kv: -kv[1][1])[:14]]
        tot_us = sum(v[1] for v in groups.values())
        out["gemm_time_weighted_mhz"] = round(sum(v[2] / v[0] * v[1] for v in groups.values()) / tot_us) if tot_us else None
    print("ARM " + json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--pin", type=int, default=0, help="MHz for the pinned-clock repetition (0: skip)")
    ap.add_argument("--dtype", default="bf16r")
    ap.add_argument("--arm", action="store_true")
    args = ap.parse_args()
    if args.arm:
        return arm(args)
    clk_lib = os.path.join(ROOT, "abl", "libdic_clk.so")
    have_probe = os.path.exists(clk_lib)
    print(f"# power A/B of the training step (B=512, 12 layers, {args.dtype}); {args.steps} steps per arm; clock-probe build: {'yes' if have_probe else 'NO (abl/libdic_clk.so missing: no per-launch clocks)'}")
    print("# idle:", " ".join(re.findall(r"(?:Power \(W\)|sclk clock level): [^\n]*", smi(["--showpower", "--showclocks"]))))

    def run_arms(label):
        res = {}
        for name, opts in (("8-wave kernel everywhere", "gemm_w4a=0,sample_w4a=0"), ("four-wave asm kernel, every eligible launch", "gemm_w4a=1,gemm_w4a_mask=0xff"),
                           ("8-wave kernel everywhere, again", "gemm_w4a=0,sample_w4a=0"), ("four-wave asm kernel, again", "gemm_w4a=1,gemm_w4a_mask=0xff")):
            env = dict(os.environ, DIC_OPTIONS=opts)
            if have_probe:
                env["DIC_HIP_LIB"] = clk_lib
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--arm", "--steps", str(args.steps), "--dtype", args.dtype], env=env, capture_output=True, text=True, timeout=900)
            m = re.search(r"^ARM (.*)$", r.stdout, re.M)
            if not m:
                print(f"## {label}: {name}: FAILED\n{(r.stdout + r.stderr)[-1500:]}")
                continue
            d = json.loads(m.group(1))
            res[name] = d
            print(f"## {label}: {name}: {d['ms_per_step']} ms/step, {d['power_w']} W mean (max {d['power_w_max']}, cap {d['cap_w']}), rocm-smi sclk {d['sclk_mhz']} MHz, "
                  f"{d['joules_per_step']} J/step" + (f", GEMM-time-weighted in-kernel clock {d['gemm_time_weighted_mhz']} MHz" if d.get("gemm_time_weighted_mhz") else ""))
            for k in d.get("kernels", []):
                print(f"      tag {k['tag']:5d}  {k['M']:6d} x {k['N']:6d} x {k['K']:6d}  {k['per_step']:5.1f}/step  {k['us']:8.1f} us  {k['mhz_in_kernel']:5d} MHz in kernel")
        return res

    run_arms("free clock")
    if args.pin:
        accepted = None
        for how, set_, reset in (("--setperfdeterminism", ["--setperfdeterminism", str(args.pin)], ["--resetperfdeterminism"]),
                                 ("--setsclk", ["--setperflevel", "manual", "--setsclk", "1"], ["--setperflevel", "auto"]),
                                 ("--setpoweroverdrive", ["--setpoweroverdrive", "1000", "--autorespond", "y"], ["--resetpoweroverdrive", "--autorespond", "y"])):
            out = smi(set_)
            ok = not re.search(r"(not supported|fail|error|unable|denied|invalid)", out, re.I) and out.strip() != ""
            print(f"# rocm-smi {' '.join(set_)}: {'accepted' if ok else 'REFUSED'}: {' | '.join(l.strip() for l in out.splitlines() if l.strip() and '===' not in l)[:300]}")
            if ok:
                accepted = (how, reset)
                break
        if accepted:
            try:
                run_arms(f"pinned ({accepted[0]} {args.pin})")
            finally:
                print("# reset:", " | ".join(l.strip() for l in smi(accepted[1]).splitlines() if l.strip() and "===" not in l)[:200])
        else:
            print("# no clock / power pin was accepted on this box: the pinned-clock arm (b) could not be run here")


if __name__ == "__main__":
    main()

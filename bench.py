#!/usr/bin/env python3
"""Headline benchmark: training captions/sec of the CLIP-DDPM hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One "step" = one `train_func` call: embed -> q_sample -> ONE stacked encoder pass over the x_t and x_1 rows ->
embedding + rounding losses -> backward -> (RCCL all-reduce) -> AdamW, on a synthetic batch already resident in HBM.
Workload (BASELINE.json configs[1]): per-GPU batch 512 captions, seq_len 16 (+2 CLIP rows), 12-layer bert-base-width
denoiser ("bert-base" of the north_star; the reference's own depth is 6 -- `--layers 6`), concat fusion, linear
beta schedule T=100, bf16 operands with fp32 accumulation/statistics/optimizer, dropout 0.1 as the reference trains,
SAMPLE_SIZE S=1 timestep per caption per step (the reference default S=100 is `--sample-size 100 --batch 8`).
Weak scaling: per-GPU work is fixed; value = N * 512 * K / max-over-ranks time.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic GFLOP per denoiser sequence, forward+backward (lm_head backward = dX only)
def gflop_per_seq(L, layers):
    T, D, F, V = L + 2, 768, 3072, 30522
    per_layer = 2 * T * D * (3 * D + D + 2 * F) + 2 * 2 * T * T * D
    fwd = layers * per_layer + 2 * T * D * D + 2 * 2 * T * D * 512 + 2 * L * D * V
    bwd = 2 * (fwd - 2 * L * D * V) + 2 * L * D * V
    return (fwd + bwd) / 1e9


def relaunch(n):
    """Re-exec this command under torch.distributed.run with n ranks on this node; never returns."""
    import socket
    import subprocess
    try:
        import torch
        have = torch.cuda.device_count()
    except Exception:
        have = 0
    if have < n:
        sys.exit(f"bench.py: --gpus {n} needs {n} visible GPUs, this node shows {have}")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="captions per GPU per step")
    ap.add_argument("--sample-size", type=int, default=1, help="SAMPLE_SIZE S: noised copies per caption per step")
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--seq-len", type=int, default=16)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    # --gpus N without a launcher: start the N ranks ourselves (one process per GPU, RCCL over xGMI) exactly as the driver would
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch(args.gpus)
    import torch
    dic = importlib.import_module("diffusion-image-captioning_amd")
    rank, world, local = dic.parallel.init_from_env()
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the job has WORLD_SIZE={world}: refusing to report a line for a different GPU count")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    B, S, L = args.batch, args.sample_size, args.seq_len
    dic.cfg.update(BATCH_SIZE=B, SAMPLE_SIZE=S, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, VOCAB_SIZE=30522,
                   CLASSIFIER_FREE_WEIGHT=0.0, CLIP_ADDING_METHOD="concat", LOSS_FUNC="series_sum_sample_mean",
                   X_0_PREDICTION=True, ROUNDING_WEIGHT=0.5)
    E = dic.synth.vocab_embedding(30522, 768, 0)
    model = dic.DistilBertModel(E, E, config=dict(n_layers=args.layers, dropout=0.1, attention_dropout=0.1), dtype=args.dtype,
                                device=dev, seed=0)            # same random init on every rank (DDP invariant)
    dic.parallel.configure_model_for_rank(model)
    trainer = dic.AdamW(model.parameters(), lr=1e-4)
    x = {k: torch.from_numpy(v).to(dev) for k, v in dic.synth.batch(B, L, 30522, seed=1 + rank).items()}
    dic.seed_noise(dic.parallel.rank_seed(1234))          # every rank its own eps stream (configure_model_for_rank did dropout + guidance)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        dic.train_func(model, trainer, x)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = dic.train_func(model, trainer, x)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax)
    loss_val = float(out[0])
    captions = world * B * args.steps
    value = captions / dt

    # ---- roofline leg: the same K steps again with every GEMM launch bracketed by hipEvents on its stream
    roof = None
    if not args.no_roofline:
        Lh = dic.lib()
        model.wgrad_stream_enabled = False      # serial launches for this leg: a kernel's duration is only meaningful when it runs alone
        Lh.dic_prof_begin(args.steps * (args.layers * 16 + 64))
        for _ in range(args.steps):
            dic.train_func(model, trainer, x)
        torch.cuda.synchronize()
        ms, fl, n = C.c_double(), C.c_double(), C.c_int()
        Lh.dic_prof_end(C.byref(ms), C.byref(fl), C.byref(n))
        model.wgrad_stream_enabled = True
        peak = 2500.0 if args.dtype == "bf16" else 157.3
        ach = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
        # HBM traffic of the same kernels comes from PMC passes (rocprofv3 --pmc cannot run inside this process): the committed
        # summary of the last collection, valid for the default workload only
        traffic, traffic_src = None, None
        pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_v7_pmc_gemm_traffic.json")
        if os.path.exists(pmc) and (B, S, L, args.layers, args.dtype) == (512, 1, 16, 12, "bf16"):
            with open(pmc) as fh:
                pj = json.load(fh)
            traffic = round(pj["gemm_fetch_bytes_per_launch_x2"] + pj["gemm_write_bytes_per_launch"])
            traffic_src = "bytes/launch averaged over the step's GEMM launches, FETCH_SIZE(x2)+WRITE_SIZE, profiles/r01_v7_pmc_gemm_traffic.json"
        roof = {"bound": "mfma", "kernel": "gemm_kernel (all layouts/epilogues)", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": traffic, "traffic_note": traffic_src, "launches_per_step": n.value // max(args.steps, 1),
                "gemm_ms_per_step": round(ms.value / args.steps, 3), "gemm_gflop_per_step": round(fl.value / args.steps / 1e9, 1)}
    barrier()

    # ---- CPU baseline: the oracle (a port of the reference step) on this host's cores, bounded sample, rank 0 at N=1 only
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ref_model as R
        cb = 16
        rcfg = R.Config(BATCH_SIZE=cb, SAMPLE_SIZE=S, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, n_layers=args.layers, vocab=30522)
        om = R.build(rcfg, dic.synth.denoiser_state(args.layers, 0), E)
        otr = R.AdamW(om.parameters(), lr=1e-4)
        xb = {k: torch.from_numpy(v) for k, v in dic.synth.batch(cb, L, 30522, 1).items()}
        times = []
        for i in range(3):
            t = torch.from_numpy(dic.synth.timesteps(S, 100, i))
            nz = [torch.randn(cb, L, 768) for _ in range(2)]
            c0 = time.perf_counter()
            R.train_func(om, otr, xb, t=t, noises=nz)
            times.append(time.perf_counter() - c0)
        step = sorted(times[1:])[0] if len(times) > 1 else times[0]
        cpu = {"value": round(cb / step, 3), "unit": "captions/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"oracle/ref_model.py train_func, {cb} captions/step (S={S}, {args.layers} layers, fp32), best of 2 steps after 1 warm-up"}

    if rank == 0:
        gf = gflop_per_seq(L, args.layers) * (S + 1)
        line = {
            "metric": "training captions/sec (seq16, bert-base)", "value": round(value, 2), "unit": "captions/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"train_func: B={B}/GPU x S={S} (+x_1 pass) = {(S + 1) * B} sequences x {L}+2 tokens (the unguided text row is "
                                   f"skipped), {args.layers}-layer DistilBERT-width denoiser, concat fusion, linear beta T=100, dropout 0.1, AdamW",
                       "global_batch": world * B, "seq_len": L, "sample_size": S, "n_layers": args.layers,
                       "parallelism": f"dp{world}", "loss": round(loss_val, 4),
                       "algorithmic_tflop_per_s": round(value * gf / 1e3, 2)},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

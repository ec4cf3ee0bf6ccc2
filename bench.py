#!/usr/bin/env python3
"""Headline benchmark: training captions/sec of the CLIP-DDPM hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (N > 1 without a launcher: bench.py starts its own N ranks)
    python bench.py --mode sample                        (BASELINE.json config 4 as the headline line instead)

One "step" = one `train_func` call: embed -> q_sample -> ONE stacked encoder pass over the x_t and x_1 rows ->
embedding + rounding losses -> backward -> (RCCL all-reduce) -> AdamW, on a synthetic batch already resident in HBM.
Workload (BASELINE.json configs[1]): per-GPU batch 512 captions, seq_len 16 (+2 CLIP rows), 12-layer bert-base-width
denoiser ("bert-base" of the north_star; the reference's own depth is 6 -- `--layers 6`), concat fusion, linear
beta schedule T=100, bf16 operands with fp32 accumulation/statistics/optimizer, dropout 0.1 as the reference trains,
SAMPLE_SIZE S=1 timestep per caption per step (the reference default S=100 is `--sample-size 100 --batch 8`).
Weak scaling: per-GPU work is fixed; value = N * 512 * K / max-over-ranks time.

Besides the contract's keys the line carries (rank 0, N = 1): `roofline` (GEMM launches bracketed by hipEvents), `cpu_baseline` (the CPU
port on a bounded sample of the same workload, median of 3) and `cpu_baseline_config1` (the BASELINE.json configs[0] shape: B=8, S=100,
6 layers), `bf16_vs_fp32_loss_rel` (the same eval step in both dtypes at the bench shape), `sampling` (configs[3]: 2048 images, 100
passes, captions/s + BLEU-4 of the bf16 ids against the fp32 ids) and `seq32_cfg` (configs[4] on one GPU: seq_len 32 + guidance).
`--quick` skips those extras.
"""
import argparse
import ctypes as C
import glob
import hashlib
import importlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


# algorithmic GFLOP per denoiser sequence, forward+backward (lm_head backward = dX only)
def gflop_per_seq(L, layers, T=None):
    """T: tokens per sequence (default L + 2 = the reference's concat sequence; L + 1 when the unguided text row is skipped)."""
    T, D, F, V = (L + 2 if T is None else T), 768, 3072, 30522
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
    if have < n and os.environ.get("DIC_DIST_SHARE_GPU", "0") != "1" and "--dry-run" not in sys.argv:   # (the shared-GPU gloo rig of the tests puts every rank on GPU 0)
        sys.exit(f"bench.py: --gpus {n} needs {n} visible GPUs, this node shows {have}")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def csrc_sha():
    """Identity of the kernel sources a PMC traffic file must have been collected on to be quoted."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "diffusion-image-captioning_amd", "csrc")
    # every text the compiler reads: *.hip, *.h and the generated *.inc (the asm GEMM's bodies are in gemm_w4a_asm.inc)
    for f in sorted(glob.glob(os.path.join(d, "*.h*")) + glob.glob(os.path.join(d, "*.inc"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic():
    """HBM bytes per GEMM launch from the committed PMC collection -- only if it was taken on exactly these kernel sources."""
    sha = csrc_sha()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_gemm_traffic.json")), reverse=True):
        try:
            pj = json.load(open(f))
        except Exception:
            continue
        if pj.get("csrc_sha") == sha:
            return round(pj["gemm_fetch_bytes_per_launch_x2"] + pj["gemm_write_bytes_per_launch"]), \
                f"bytes/launch averaged over the step's GEMM launches, FETCH_SIZE(x2)+WRITE_SIZE, profiles/{os.path.basename(f)} (csrc {sha})", pj
    return None, f"no PMC collection for csrc {sha} under profiles/ (rocprofv3 --pmc needs its own passes: scripts/pmc_traffic.sh)", None


class PowerSampler:
    """rocm-smi samples (socket power, shader clock) on a host thread while a leg runs: whether the step sits at the package power limit is part
    of reading its roofline fraction (round 4: it does -- 1.33-1.35 kW of 1.4 kW, sclk ~2.27 GHz instead of 2.4; profiles/r04_power_probe.txt)."""

    def __init__(self, device_index=0, period=0.7):
        import threading
        self.dev, self.period, self.rows, self.cap = device_index, period, [], None
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)

    @staticmethod
    def _smi(args):
        import re
        import subprocess
        try:
            out = subprocess.run(["rocm-smi"] + args, capture_output=True, text=True, timeout=10).stdout
        except Exception:
            return None, None, None
        pw = re.search(r"Socket Graphics Package Power \(W\): ([0-9.]+)", out) or re.search(r"Average Graphics Package Power \(W\): ([0-9.]+)", out)
        ck = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
        cap = re.search(r"Max Graphics Package Power \(W\): ([0-9.]+)", out)
        return (float(pw.group(1)) if pw else None, float(ck.group(1)) if ck else None, float(cap.group(1)) if cap else None)

    def _run(self):
        while not self._stop.is_set():
            pw, ck, _ = self._smi(["-d", str(self.dev), "--showpower", "--showclocks"])
            if pw is not None or ck is not None:
                self.rows.append((pw, ck))
            self._stop.wait(self.period)

    def __enter__(self):
        self.cap = self._smi(["-d", str(self.dev), "--showmaxpower"])[2]
        self._th.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        self._th.join(timeout=15)

    def summary(self):
        pw = [r[0] for r in self.rows if r[0] is not None]
        ck = [r[1] for r in self.rows if r[1] is not None]
        if not pw and not ck:
            return None
        return {"samples": len(self.rows), "socket_power_w_mean": round(sum(pw) / len(pw), 1) if pw else None, "socket_power_w_max": max(pw) if pw else None,
                "power_cap_w": self.cap, "sclk_mhz_mean": round(sum(ck) / len(ck), 1) if ck else None, "sclk_mhz_min": min(ck) if ck else None,
                "source": "rocm-smi --showpower --showclocks on a host thread during the sustained leg"}


def sampling_leg(dic, torch, E, dev, batch, passes, layers, dtype, reps=2, bleu_batch=256, oracle_captions=64):
    """BASELINE.json configs[3]: x0-prediction sampling loop (ref :611-621), logits/argmax only after the last pass."""
    dic.cfg.update(MAX_LENGTH=16, CLASSIFIER_FREE_WEIGHT=0.0, CLIP_ADDING_METHOD="concat", VOCAB_SIZE=30522)
    model = dic.DistilBertModel(E, E, config=dict(n_layers=layers), dtype=dtype, device=dev)
    model.eval()
    img = torch.from_numpy(dic.synth.batch(batch, 16, 30522, 2)["image_clip"]).to(dev)
    dic.sample(model, img, steps=2)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        dic.sample(model, img, steps=passes)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    out = {"metric": "sampling captions/sec", "value": round(batch / best, 1), "unit": "captions/s", "batch": batch, "denoising_passes": passes,
           "n_layers": layers, "dtype": dtype, "ms_per_pass": round(best / passes * 1e3, 3)}
    # forward-GEMM roofline of the denoising passes: GEMM launches of an 8-pass loop minus those of a 2-pass loop (the difference is six pure
    # passes: the one-off CLIP projection and the exact-fp32 rounding head drop out), launch by launch (no graph replay: the per-launch events
    # need real launches)
    Lh = dic.lib()
    opt = importlib.import_module("diffusion-image-captioning_amd.options").OPT
    keep_graph, opt.sample_graph = opt.sample_graph, False
    acc = {}
    for npass in (8, 2):
        Lh.dic_prof_begin(npass * (layers * 6 + 16))
        dic.sample(model, img, steps=npass)
        torch.cuda.synchronize()
        ms, fl, n = C.c_double(), C.c_double(), C.c_int()
        Lh.dic_prof_end(C.byref(ms), C.byref(fl), C.byref(n))
        acc[npass] = (ms.value, fl.value, n.value)
    opt.sample_graph = keep_graph
    dms, dfl, dn = (acc[8][k] - acc[2][k] for k in range(3))
    if dms > 0:
        peak = 2500.0 if dtype != "fp32" else 157.3
        ach = dfl / (dms * 1e-3) / 1e12
        out["roofline"] = {"bound": "mfma", "kernel": "forward GEMMs of one denoising pass (QKV, out-proj, FFN1+GELU, FFN2 per layer + the MLM-head transform)",
                           "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "gemm_ms_per_pass": round(dms / 6, 3),
                           "launches_per_pass": dn // 6}
    if dtype in ("bf16", "bf16r") and bleu_batch > 0:
        # BLEU-4 of the bf16 loop's ids against the fp32 loop's ids from the SAME start noise (the fp32 path is the one the -m gpu tests
        # pin bit-exactly to the reference's ids on the golden fixture): how far the bf16 passes drift in token space
        m32 = dic.DistilBertModel(E, E, config=dict(n_layers=layers), dtype="fp32", device=dev)
        m32.load_state_dict(model.state_dict())
        m32.eval()
        start = torch.randn(bleu_batch, 18, 768, generator=torch.Generator().manual_seed(11)).to(dev)
        ids16 = dic.sample(model, img[:bleu_batch], steps=passes, start=start).cpu()
        ids32 = dic.sample(m32, img[:bleu_batch], steps=passes, start=start).cpu()
        out["bleu4_bf16_vs_fp32_ids"] = round(dic.bleu.corpus_bleu([r.tolist() for r in ids16], [[r.tolist()] for r in ids32]), 4)
        out["token_agreement"] = round(float((ids16 == ids32).float().mean()), 4)
        out["bleu_captions"] = bleu_batch
        if oracle_captions > 0:
            # BASELINE configs[3] "BLEU-4 vs ref": the SAME loop run by the CPU oracle (oracle/ref_model.py::sample -- the checker, on the host cores)
            # from the same start noise, on the first `oracle_captions` images; agreement bucketed by the oracle's own top-1 / top-2 logit margin
            from oracle import ref_model as R
            nb = min(oracle_captions, bleu_batch)
            c0 = time.perf_counter()
            om = R.build(R.Config(MAX_LENGTH=16, n_layers=layers, vocab=30522), {k: v.cpu().numpy() for k, v in model.state_dict().items()}, E, requires_grad=False)
            oids, ohid = R.sample(om, img[:nb].cpu(), steps=passes, start=start[:nb].cpu())
            lg = ohid[:, :16].double() @ torch.from_numpy(E).double().t()
            top2 = lg.topk(2, -1).values
            margin = (top2[..., 0] - top2[..., 1])
            ref = [[r.tolist()] for r in oids]
            vs = {"captions": nb, "passes": passes, "oracle_seconds": round(time.perf_counter() - c0, 1)}
            for name, ids in (("fp32", ids32[:nb]), ("bf16", ids16[:nb])):
                same = ids == oids
                vs[f"bleu4_{name}_vs_oracle_ids"] = round(dic.bleu.corpus_bleu([r.tolist() for r in ids], ref), 4)
                vs[f"token_agreement_{name}"] = round(float(same.float().mean()), 4)
                vs[f"agreement_by_oracle_margin_{name}"] = {f"[{lo},{hi})": [int(((margin >= lo) & (margin < hi)).sum()), round(float(same[(margin >= lo) & (margin < hi)].float().mean()), 4)]
                                                            for lo, hi in ((0, 0.01), (0.01, 0.03), (0.03, 0.1), (0.1, 1e9)) if bool(((margin >= lo) & (margin < hi)).any())}
            out["vs_oracle"] = vs
            del om
        del m32
    del model
    torch.cuda.empty_cache()
    return out


def cpu_leg(dic, torch, E, cb, S, L, layers, steps=3):
    """The CPU port of the reference step (oracle/ref_model.py) on this host's cores: median of `steps` after one warm-up."""
    from oracle import ref_model as R
    rcfg = R.Config(BATCH_SIZE=cb, SAMPLE_SIZE=S, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, n_layers=layers, vocab=30522)
    om = R.build(rcfg, dic.synth.denoiser_state(layers, 0), E)
    otr = R.AdamW(om.parameters(), lr=1e-4)
    xb = {k: torch.from_numpy(v) for k, v in dic.synth.batch(cb, L, 30522, 1).items()}
    times = []
    for i in range(steps + 1):
        t = torch.from_numpy(dic.synth.timesteps(S, 100, i))
        nz = [torch.randn(cb, L, 768) for _ in range(2)]
        c0 = time.perf_counter()
        R.train_func(om, otr, xb, t=t, noises=nz)
        times.append(time.perf_counter() - c0)
    med = statistics.median(times[1:])
    return {"value": round(cb / med, 3), "unit": "captions/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/ref_model.py train_func, {cb} captions/step (S={S}, L={L}, {layers} layers, fp32, linear T=100), median of {steps} steps "
                      f"after 1 warm-up ({med:.2f} s/step)"}


def dry_run(args):
    """`--dry-run`: the plumbing of an N-rank run end to end WITHOUT a GPU and without a kernel -- what has to work the first time the driver starts
    `bench.py --gpus 8` on a real node, rehearsed where it can be: the relauncher, torchrun's environment, `parallel.init_from_env` (gloo over CPU
    tensors here, RCCL there), per-rank seeds with a shared timestep stream, the real `GradReducer` over a flat 12-layer gradient buffer (sliced
    schedule or options.dp_single), barrier + max-over-ranks timing, the `data_parallel` block and the JSON schema.  The "step" only fills the
    gradient buffer with a rank-dependent pattern and checks the reduced values; the line says so (`dry_run`, `data`) and its value is meaningless."""
    import torch
    dic = importlib.import_module("diffusion-image-captioning_amd")
    OPT = importlib.import_module("diffusion-image-captioning_amd.options").OPT
    ParamStore = importlib.import_module("diffusion-image-captioning_amd.params").ParamStore
    diffusion = importlib.import_module("diffusion-image-captioning_amd.diffusion")
    rank, world, local = dic.parallel.init_from_env()
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the job has WORLD_SIZE={world}")
    store = ParamStore(args.layers, "cpu", bf16_shadow=False)

    class Model:                                   # what configure_model_for_rank and GradReducer touch on a Denoiser
        params, ops, device, dropout_seed_base, rank_rows_forced = store, None, torch.device("cpu"), 0x5EED0000, True

        def set_dropout_seed(self, s_):
            self.seed = s_
    model = Model()
    dic.parallel.configure_model_for_rank(model)
    dic.seed_noise(dic.parallel.rank_seed(1234))
    t_seeds = [diffusion._next_t_seed() for _ in range(3)]
    dic.parallel.assert_shared_timestep_seed()
    seeds = [None] * world
    torch.distributed.all_gather_object(seeds, (model.seed, diffusion._state["noise_seed"], t_seeds))
    assert len({s_[0] for s_ in seeds}) == world and len({s_[1] for s_ in seeds}) == world and len({tuple(s_[2]) for s_ in seeds}) == 1, seeds
    B = args.batch or 512

    class Trainer:
        grad_scale = 1.0

    def step():
        store.G.fill_(float(rank + 1))
        red = dic.parallel.GradReducer(model)
        for i in reversed(range(args.layers)):
            red.layer_done(i)
        tr = Trainer()
        red.finish(tr)
        want = world * (world + 1) / 2
        assert tr.grad_scale == 1.0 / world and float(store.G.min()) == want == float(store.G.max()), (float(store.G.min()), want)
        return red
    for _ in range(args.warmup):
        step()
    torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        red = step()
    torch.distributed.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt])
    torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    mine = torch.tensor([B * args.steps / dt])
    allr = [torch.zeros_like(mine) for _ in range(world)]
    torch.distributed.all_gather(allr, mine)
    dp_info = {"rccl_ranks": torch.distributed.get_world_size(), "backend": torch.distributed.get_backend(), "rccl_version": None,
               "devices_by_rank": ["cpu"] * world, "collectives_per_step": red.n_collectives,
               "mode": "single all-reduce after the backward (options.dp_single)" if OPT.dp_single else f"slices of {OPT.dp_group} layers issued from the backward + tail",
               "allreduce_ms_per_step": None, "gradient_bytes": int(store.numel) * 4, "per_rank_captions_per_s": [round(float(v), 1) for v in allr]}
    if rank == 0:
        print(json.dumps({"metric": "training captions/sec (seq16, bert-base)", "value": round(world * B * args.steps / float(tmax), 2), "unit": "captions/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(float(tmax) / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "none", "data": "DRY RUN: no kernel ran; CPU tensors over gloo, the step only fills and exchanges the gradient buffer",
                          "dry_run": True, "config": {"workload": "plumbing rehearsal of bench.py --gpus N (relaunch, rendezvous, rank seeds, gradient exchange schedule, "
                                                                  "timing, schema)", "global_batch": world * B, "n_layers": args.layers, "parallelism": f"dp{world}"},
                          "roofline": None, "cpu_baseline": None, "data_parallel": dp_info, "options_non_default": OPT.non_default()}), flush=True)
    torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", default="train", choices=["train", "sample"])
    ap.add_argument("--batch", type=int, default=None, help="captions per GPU per step (train: 512, sample: 2048)")
    ap.add_argument("--sample-size", type=int, default=1, help="SAMPLE_SIZE S: noised copies per caption per step")
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--seq-len", type=int, default=16)
    ap.add_argument("--cfg-weight", type=float, default=0.0, help="classifier-free guidance weight (configs[4]: 0.3 with --seq-len 32)")
    ap.add_argument("--passes", type=int, default=100, help="denoising passes of --mode sample")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "bf16r", "bf16w", "bf16m", "fp32"],
                    help="bf16 (default; 'bf16m' is its older name): the engine inside north_star's 1e-4 loss tolerance -- mean-row lo-weight correction of "
                         "every forward Linear + centred bf16 residual stream; bf16r: the raw bf16 engine without them (the default line carries it as "
                         "throughput_mode); bf16w: the default's exact form, the lo weight halves as a second K-loop pass")
    ap.add_argument("--dry-run", action="store_true", help="rehearse the multi-rank plumbing without a GPU (gloo, CPU tensors, no kernels): see dry_run()")
    ap.add_argument("--no-oracle-leg", action="store_true", help="skip loss_rel_vs_cpu_oracle (one forward of the CPU oracle at the bench shape, ~1-2 min)")
    ap.add_argument("--sustained", type=int, default=500, help="steps of the extra sustained leg of the default line (0: skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--quick", action="store_true", help="headline + roofline only (no CPU legs, no sampling / seq32 / dtype-delta extras)")
    args = ap.parse_args()
    if args.dtype == "bf16m":
        args.dtype = "bf16"

    # --gpus N without a launcher: start the N ranks ourselves (one process per GPU, RCCL over xGMI) exactly as the driver would
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch(args.gpus)
    if args.dry_run:
        return dry_run(args)
    import torch
    dic = importlib.import_module("diffusion-image-captioning_amd")
    OPT = importlib.import_module("diffusion-image-captioning_amd.options").OPT
    rank, world, local = dic.parallel.init_from_env()
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the job has WORLD_SIZE={world}: refusing to report a line for a different GPU count")
    if world > 1:
        # a multi-GPU line is an RCCL-over-xGMI line: any other backend must have been asked for by name (DIC_DIST_BACKEND, the shared-GPU
        # gloo rig of the tests) and is labelled as such in `data_parallel`
        be = torch.distributed.get_backend()
        if be != "nccl" and not os.environ.get("DIC_DIST_BACKEND"):
            sys.exit(f"bench.py: --gpus {args.gpus} initialised the '{be}' backend, not nccl (= RCCL on ROCm): refusing to report a multi-GPU number over it")
        if torch.distributed.get_world_size() != args.gpus:
            sys.exit(f"bench.py: the communicator has {torch.distributed.get_world_size()} ranks, --gpus says {args.gpus}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    E = dic.synth.vocab_embedding(30522, 768, 0)

    if args.mode == "sample":
        line = sampling_leg(dic, torch, E, dev, args.batch or 2048, args.passes, args.layers, args.dtype, reps=3)
        line.update({"n_gpus": world, "steps": args.passes, "warmup": 2, "ms_per_step": line["ms_per_pass"], "higher_is_better": True, "scaling": "weak",
                     "vs_baseline": None, "data": "synthetic",
                     "config": {"workload": f"sample(): {line['batch']} images, {args.passes} x0-prediction passes of the {args.layers}-layer denoiser, "
                                            f"rounding (argmax over 30522) after the last pass"}})
        if rank == 0:
            print(json.dumps(line))
        return

    B, S, L = args.batch or 512, args.sample_size, args.seq_len
    w = args.cfg_weight

    def configure(L_, w_):
        dic.cfg.update(BATCH_SIZE=B, SAMPLE_SIZE=S, MAX_LENGTH=L_, STEP_TOT=100, COSIN_SCHEDULE=False, VOCAB_SIZE=30522,
                       CLASSIFIER_FREE_WEIGHT=w_, CLASSIFIER_FREE_PROB=0.2, CLIP_ADDING_METHOD="concat", LOSS_FUNC="series_sum_sample_mean",
                       X_0_PREDICTION=True, ROUNDING_WEIGHT=0.5)
    configure(L, w)
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

    # DIC_STEP_GRAPH=1 (one GPU, no guidance): the step is captured once as a hipGraph and replayed (diffusion-image-captioning_amd/graph.py: same
    # kernels, same arithmetic, bit-identical to the eager step).  Measured in round 3: 15.94 ms per replayed step against 15.42 ms for the eager
    # launches on the same box (profiles/r03_step_graph_ab.txt) -- the two-stream overlap of the eager step is better than what the graph's
    # branch scheduling gives -- so the eager step stays the default.
    step = lambda: dic.train_func(model, trainer, x)
    graph_note = "eager launches"
    if world == 1 and w <= 0 and os.environ.get("DIC_STEP_GRAPH", "0") == "1":
        try:
            step = dic.GraphedTrainStep(model, trainer, x, warmup=2)
            graph_note = "hipGraph replay of the captured step"
        except Exception as e:                       # capture refused: time the eager step
            graph_note = f"eager launches (graph capture failed: {type(e).__name__}: {e})"[:200]
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    dt_local = dt
    if world > 1:
        tmax = torch.tensor([dt], device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax)
    loss_val = float(out[0])
    captions = world * B * args.steps
    value = captions / dt
    dp_info = None
    if world > 1:
        # per-rank rates and the exchange's own time (a few extra steps with events around every collective, DIC_DP_TIMING)
        mine = torch.tensor([B * args.steps / dt_local], device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allr, mine)
        OPT.dp_timing = True
        ar_ms, ncoll = [], 0
        for _ in range(3):
            dic.train_func(model, trainer, x)
            red = dic.parallel.GradReducer.last
            if red is not None:
                v = red.allreduce_ms()
                if v is not None:
                    ar_ms.append(v)
                ncoll = red.n_collectives
        OPT.dp_timing = False
        barrier()
        try:
            rccl_ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            rccl_ver = None
        devs = [None] * world
        torch.distributed.all_gather_object(devs, f"{torch.cuda.current_device()}:{torch.cuda.get_device_properties(torch.cuda.current_device()).name}")
        dp_info = {"rccl_ranks": torch.distributed.get_world_size(), "backend": torch.distributed.get_backend(), "rccl_version": rccl_ver,
                   "devices_by_rank": devs, "collectives_per_step": ncoll,
                   "mode": "single all-reduce after the backward (options.dp_single)" if OPT.dp_single else
                           f"slices of {OPT.dp_group} layers issued from the backward + tail",
                   "allreduce_ms_per_step": round(sum(ar_ms) / len(ar_ms), 3) if ar_ms else None,
                   "allreduce_note": "sum over the step's collectives of issue -> completion on the compute stream, rank 0 (they overlap the backward)",
                   "gradient_bytes": int(model.params.numel) * 4, "per_rank_captions_per_s": [round(float(v), 1) for v in allr]}

    # ---- sustained leg: the same step for --sustained more steps (the 20-step headline is 0.3 s of GPU time; this one says what a run holds)
    leg_errors = {}
    sustained = None
    if world == 1 and rank == 0 and args.sustained > 0 and not args.quick:
        try:
            torch.cuda.synchronize()
            with PowerSampler(local) as ps:
                c0 = time.perf_counter()
                for _ in range(args.sustained):
                    step()
                torch.cuda.synchronize()
                ds = time.perf_counter() - c0
            pw_ = ps.summary()
            sustained = {"steps": args.sustained, "value": round(B * args.sustained / ds, 1), "unit": "captions/s", "ms_per_step": round(ds / args.sustained * 1e3, 3),
                         "seconds": round(ds, 2), "power": pw_}
            if pw_ and pw_.get("socket_power_w_mean"):
                # energy of one step = mean socket power x step time (what a faster-but-denser kernel has to beat when the package is power-limited)
                sustained["joules_per_step"] = round(pw_["socket_power_w_mean"] * ds / args.sustained, 2)
                sustained["captions_per_joule"] = round(B / sustained["joules_per_step"], 1)
        except Exception as e:                    # an extra leg never takes the headline line down with it
            leg_errors['sustained'] = f"{type(e).__name__}: {e}"[:400]

    # ---- roofline leg: the same K steps again with every GEMM launch bracketed by hipEvents on its stream
    roof = None
    if not args.no_roofline:
        try:
            Lh = dic.lib()
            model.wgrad_stream_enabled = False      # serial launches for this leg: a kernel's duration is only meaningful when it runs alone
            Lh.dic_prof_begin(args.steps * (args.layers * 16 + 64))
            for _ in range(args.steps):
                dic.train_func(model, trainer, x)
            torch.cuda.synchronize()
            ms, fl, n = C.c_double(), C.c_double(), C.c_int()
            alg_bytes = Lh.dic_prof_algorithmic_bytes() / max(args.steps, 1)
            Lh.dic_prof_end(C.byref(ms), C.byref(fl), C.byref(n))
            model.wgrad_stream_enabled = True
            peak = 2500.0 if args.dtype != "fp32" else 157.3
            ach = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
            traffic, traffic_src, pj = (None, "PMC traffic is collected for the default workload only", None)
            if (B, S, L, args.layers, args.dtype, w) == (512, 1, 16, 12, "bf16", 0.0):
                traffic, traffic_src, pj = pmc_traffic()
            roof = {"bound": "mfma", "kernel": "gemm_bf16_kernel (all layouts/epilogues)" if args.dtype != "fp32" else "gemm_kernel<float>", "achieved": round(ach, 2),
                    "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic, "traffic_note": traffic_src,
                    "launches_per_step": n.value // max(args.steps, 1), "gemm_ms_per_step": round(ms.value / args.steps, 3),
                    "gemm_gflop_per_step": round(fl.value / args.steps / 1e9, 1),
                    # every operand / side input / output of the step's GEMM launches once (dic_prof_algorithmic_bytes): what the PMC traffic is to be held against
                    "algorithmic_bytes_per_step": round(alg_bytes), "algorithmic_bytes_per_launch": round(alg_bytes / max(n.value // max(args.steps, 1), 1))}
            if pj is not None and "gemm_bytes_per_step" in pj:
                roof["traffic_per_step"] = round(pj["gemm_bytes_per_step"])
                roof["pmc_launches_per_step"] = pj["gemm_launches_per_step"]
                roof["fold_traffic_per_step"] = round(pj.get("fold_bytes_per_step", 0))
                roof["traffic_waste_ratio"] = round((pj["gemm_bytes_per_step"] + pj.get("fold_bytes_per_step", 0)) / max(alg_bytes, 1.0), 3)
                roof["whole_step_traffic_gb"] = round(pj["step_fetch_gb_x2"] + pj["step_write_gb"], 2)
            if args.dtype == "bf16w":
                roof["note"] = "executed flops: the forward Linears run their K loop twice (hi + lo weight halves); algorithmic flops per step are those of the bf16 line"
            if args.dtype in ("bf16", "bf16r") and w == 0.0:
                if model.ce_fused:
                    roof["logits_recompute"] = "none: the training forward of the rounding loss keeps exp(logit - c) (dic_gemm CE_EXP), every GEMM flop counted is algorithmic"
                else:
                    # DIC_CE_FUSED=0: the backward recomputes the rounding logits instead of storing them: that GEMM is executed work, not algorithmic work
                    rec = 2.0 * ((S + 1) * B * L) * 30592 * 768
                    roof["frac_excluding_logits_recompute"] = round((fl.value / args.steps - rec) / (ms.value / args.steps * 1e-3) / 1e12 / peak, 4)
        except Exception as e:                    # an extra leg never takes the headline line down with it
            model.wgrad_stream_enabled = True
            roof = None
            leg_errors['roofline'] = f"{type(e).__name__}: {e}"[:400]
    barrier()

    extras = rank == 0 and world == 1 and not args.quick
    dtype_delta = sampling = seq32 = None
    parity_fast = None
    throughput_mode = None
    if extras and args.dtype in ("bf16", "bf16r"):
        alt = "bf16r" if args.dtype == "bf16" else "bf16"         # the other of the two: the raw bf16 engine (fastest, outside the tolerance) / the default (parity) engine
        try:
            # The same eval step (same t, same noise, dropout off) in the fp32 engine (the parity dtype: within 1e-4 of the CPU reference, tests/), the
            # benchmarked bf16 engine and the parity mode "bf16m" -- at the INITIAL weights (the comparison the -m gpu tests make against the oracle)
            # and at the weights this benchmark has just trained (hundreds of AdamW steps on one synthetic batch: a degenerate state in which the denoiser
            # predicts nearly the same vector for every row, so roundings that are independent across rows at initialisation become common to all rows
            # and no longer average out of a batch mean -- profiles/r04_trained_gap_split.txt)
            from_t = torch.from_numpy(dic.synth.timesteps(S, 100, 0))
            nz = [torch.from_numpy(dic.synth.noise((B, L, 768), 3, f"eps{i}")) for i in range(2)]
            u = torch.from_numpy(dic.synth.uniform(dic.synth.stream_id("cfg", 3), (S * B, 1)))
            mk = lambda dt_: dic.DistilBertModel(E, E, config=dict(n_layers=args.layers, dropout=0.1, attention_dropout=0.1), dtype=dt_, device=dev, seed=0)

            def eval_losses(m2):
                m2.eval()
                with torch.no_grad():
                    r = dic.train_func(m2, None, x, train=False, t=from_t, noises=nz, cfg_uniform=u)
                m2.train()
                return [float(v) for v in r]
            rel = lambda got, ref: {k: round(abs(a_ - b_) / abs(b_), 8) for k, a_, b_ in zip(("total", "x_t", "x_1", "prob"), got, ref)}
            trained = model.state_dict()
            n_trained = int(trainer.t)
            m32, mw = mk("fp32"), mk(alt)
            init = m32.state_dict()                                   # (seed 0: the weights the benchmarked model started from)
            res = {}
            for tag, st_ in (("at_initial_weights", init), (f"after_{n_trained}_training_steps_on_one_batch", trained)):
                m32.load_state_dict(st_)
                mw.load_state_dict(st_)
                model.load_state_dict(st_)
                ref = eval_losses(m32)
                res[tag] = {args.dtype: rel(eval_losses(model), ref), alt: rel(eval_losses(mw), ref)}
            model.load_state_dict(trained)
            tags = list(res)
            dtype_delta = dict(res[tags[0]][args.dtype])
            dtype_delta[tags[1]] = res[tags[1]][args.dtype]
            dtype_delta["mode"] = args.dtype
            dtype_delta["note"] = ("the benchmarked mode against the fp32 HIP engine (itself within 1e-4 of the CPU oracle at this shape, tests/); first four keys: at "
                                   "the initial weights.  What separates the PLAIN bf16 engine from fp32 there is the bf16 rounding of the WEIGHTS: one perturbation "
                                   "shared by every sample, whose first-order effect a batch-mean loss does not average out (profiles/r04_weight_rounding_probe.txt); "
                                   + ("the benchmarked (default) engine adds its row-common part back (dic_lin_prep) and stores the residual stream centred on predicted "
                                      "mean rows; throughput_mode is the raw engine without either" if args.dtype == "bf16" else
                                      "this is the RAW engine; parity_fast_mode is the default engine, which adds the row-common part back (dic_lin_prep)"))
            # ALONG A TRAINING RUN (8 cycled synthetic batches, a fresh split-weight model trains; the three engines evaluate a held-out batch on its
            # weights at a few states): between the first and some hundreds of steps the denoiser's rows are nearly equal, and bf16 roundings of
            # row-common quantities no longer average out of a batch mean -- tests/test_gpu_e2e.py::test_bf16_engines_stay_near_fp32_along_a_training_run
            along, x_keep = None, x
            try:
                mt = mk("bf16")
                mt.load_state_dict(init)
                trt = dic.AdamW(mt.parameters(), lr=1e-4)
                tb = [{k: torch.from_numpy(v).to(dev) for k, v in dic.synth.batch(B, L, 30522, seed=100 + i).items()} for i in range(8)]
                held = {k: torch.from_numpy(v).to(dev) for k, v in dic.synth.batch(B, L, 30522, seed=8).items()}
                done_, along = 0, {}
                for upto in (5, 20, 80):
                    mt.train()
                    while done_ < upto:
                        dic.train_func(mt, trt, tb[done_ % 8])
                        done_ += 1
                    st_ = mt.state_dict()
                    for m_ in (m32, mw, model):
                        m_.load_state_dict(st_)
                    x = held
                    ref = eval_losses(m32)
                    along[f"after_{done_}_steps"] = {args.dtype: rel(eval_losses(model), ref), alt: rel(eval_losses(mw), ref)}
                    x = x_keep
                model.load_state_dict(trained)
                del mt, trt, tb, held
            except Exception as e:
                x = x_keep
                model.load_state_dict(trained)
                leg_errors['loss_rel_along_training'] = f"{type(e).__name__}: {e}"[:400]
            if along:
                dtype_delta["along_training_8_cycled_batches_held_out_eval"] = {k: v[args.dtype] for k, v in along.items()}
            # THE FAST MODE INSIDE north_star's 1e-4: bf16 activations and backward, hi+lo bf16 weights in the forward Linears, fp32 residual stream,
            # fp32 MLM-head pre-activation, mean-centred rounding-head input
            mw.load_state_dict(init)
            trw = dic.AdamW(mw.parameters(), lr=1e-4)
            for _ in range(3):
                dic.train_func(mw, trw, x)
            torch.cuda.synchronize()
            c0 = time.perf_counter()
            nw = 20
            for _ in range(nw):
                ow = dic.train_func(mw, trw, x)
            torch.cuda.synchronize()
            dw = (time.perf_counter() - c0) / nw
            desc = {"bf16": "bf16 (the default engine): bf16 MFMA operands / gradients; the lo halves of the fp32 master weights enter the forward Linears through "
                            "their row-common part only (mean row of the Linear's input x lo half, added to the bias: dic_lin_prep -- the part of the weights' "
                            "rounding a batch-mean loss does not average out; FFN lin1 left out), residual stream stored as bf16(value - predicted mean row) "
                            "(dic_ln_fwd_cen), fp32 MLM-head pre-activation, mean-centred rounding-head input, fp32 master weights and optimizer.  "
                            "dtype='bf16w' (the lo halves as a second K-loop pass + fp32 residual stream, ~25 % slower) gives the same distances",
                    "bf16r": "bf16r: raw bf16 weights and activations (fp32 MLM-head pre-activation, mean-centred rounding-head input), fp32 master weights and "
                             "optimizer: the fastest mode, outside north_star's 1e-4 at the initial weights (what rounds 1-4 benchmarked)"}

            def block(name, value_, ms_, steps_, loss_):
                b_ = {"dtype": desc[name], "value": round(value_, 1), "unit": "captions/s", "ms_per_step": round(ms_, 3), "steps": steps_, "loss": round(loss_, 4),
                      "loss_rel_vs_fp32": res[tags[0]][name], "loss_rel_vs_fp32_" + tags[1]: res[tags[1]][name], "tolerance": 1e-4}
                if along:
                    b_["loss_rel_vs_fp32_along_training_8_cycled_batches_held_out_eval"] = {k: v[name] for k, v in along.items()}
                return b_
            alt_block = block(alt, B / dw, dw * 1e3, nw, float(ow[0]))
            if args.dtype == "bf16":              # the benchmarked mode IS the parity mode: its block repeats the headline's numbers next to its loss distances
                parity_fast = block("bf16", value, dt / args.steps * 1e3, args.steps, loss_val)
                parity_fast["is_the_benchmarked_mode"] = True
                throughput_mode = alt_block
            else:
                parity_fast = alt_block
            del mw, trw, m32
            torch.cuda.empty_cache()
        except Exception as e:                    # an extra leg never takes the headline line down with it
            leg_errors['dtype_deltas_parity_fast_mode'] = f"{type(e).__name__}: {e}"[:400]
    fp32_mode = None
    if extras and args.dtype in ("bf16", "bf16r"):
        try:
            # the parity dtype's throughput on the same workload (fp32 MFMA peak is 1/16 of bf16's): a few steps are enough
            m32 = dic.DistilBertModel(E, E, config=dict(n_layers=args.layers, dropout=0.1, attention_dropout=0.1), dtype="fp32", device=dev, seed=0)
            tr32 = dic.AdamW(m32.parameters(), lr=1e-4)
            for _ in range(2):
                dic.train_func(m32, tr32, x)
            torch.cuda.synchronize()
            c0 = time.perf_counter()
            n32 = 5
            for _ in range(n32):
                o32 = dic.train_func(m32, tr32, x)
            torch.cuda.synchronize()
            d32 = (time.perf_counter() - c0) / n32
            fp32_mode = {"value": round(B / d32, 1), "unit": "captions/s", "ms_per_step": round(d32 * 1e3, 3), "steps": n32, "loss": round(float(o32[0]), 4),
                         "algorithmic_tflop_per_s": round(B / d32 * gflop_per_seq(L, args.layers) * (S + 1) / 1e3, 2), "mfma_f32_peak_tflops": 157.3}
            del m32, tr32
            torch.cuda.empty_cache()
            if parity_fast is not None:
                parity_fast["x_fp32_mode"] = round(parity_fast["value"] / fp32_mode["value"], 2)
        except Exception as e:                    # an extra leg never takes the headline line down with it
            leg_errors['fp32_mode'] = f"{type(e).__name__}: {e}"[:400]
    del trainer, model
    torch.cuda.empty_cache()
    if extras:
        try:
            # (sampling in the plain bf16 engine: its token ids already equal the oracle's, and the forward-only passes have nothing to average over)
            sampling = sampling_leg(dic, torch, E, dev, 2048, 100, args.layers, args.dtype)
            if (L, w) == (16, 0.0):
                # configs[4] on one GPU: seq_len 32 (+2 CLIP rows = 34 tokens: the 2x2-tile MFMA attention) with classifier-free guidance
                configure(32, 0.3)
                m5 = dic.DistilBertModel(E, E, config=dict(n_layers=args.layers, dropout=0.1, attention_dropout=0.1), dtype=args.dtype, device=dev, seed=0)
                tr5 = dic.AdamW(m5.parameters(), lr=1e-4)
                x5 = {k: torch.from_numpy(v).to(dev) for k, v in dic.synth.batch(B, 32, 30522, seed=1).items()}
                for _ in range(3):
                    dic.train_func(m5, tr5, x5)
                torch.cuda.synchronize()
                c0 = time.perf_counter()
                n5 = 8
                for _ in range(n5):
                    o5 = dic.train_func(m5, tr5, x5)
                torch.cuda.synchronize()
                d5 = (time.perf_counter() - c0) / n5
                seq32 = {"metric": "training captions/sec (seq32 + classifier-free guidance w=0.3, p=0.2)", "value": round(B / d5, 1), "unit": "captions/s",
                         "ms_per_step": round(d5 * 1e3, 3), "steps": n5, "batch": B, "seq_len": 32, "n_layers": args.layers, "dtype": args.dtype,
                         "sequences_per_step": f"{S * B} x_t + ~{0.8 * S * B:.0f} guided copies + {B} x_1, 34 tokens each", "loss": round(float(o5[0]), 4)}
                del m5, tr5
                torch.cuda.empty_cache()
                configure(L, w)
        except Exception as e:                    # an extra leg never takes the headline line down with it
            leg_errors['sampling_seq32'] = f"{type(e).__name__}: {e}"[:400]

    # ---- the benchmarked dtype against the CPU ORACLE at the bench shape (forward only: one evaluation of oracle/ref_model.py on the host cores -- the
    # checker, outside every timed region), and the reference's own default shape (SAMPLE_SIZE 100, batch 8, 6 layers) on the GPU
    oracle_rel = ref_faithful = None
    if extras and not args.no_oracle_leg and args.dtype != "fp32" and (L, w, S) == (16, 0.0, 1):
        try:
            from oracle import ref_model as R
            st0 = dic.synth.denoiser_state(args.layers, 0)
            xb0 = dic.synth.batch(B, L, 30522, 1)
            t0_ = torch.from_numpy(dic.synth.timesteps(S, 100, 0))
            nz0 = [torch.from_numpy(dic.synth.noise((B, L, 768), 3, f"eps{i}")) for i in range(2)]
            c0 = time.perf_counter()
            om = R.build(R.Config(BATCH_SIZE=B, SAMPLE_SIZE=S, MAX_LENGTH=L, STEP_TOT=100, COSIN_SCHEDULE=False, n_layers=args.layers, vocab=30522), st0, E,
                         requires_grad=False)
            with torch.no_grad():
                oref = [float(v) for v in R.train_func(om, None, {k: torch.from_numpy(v) for k, v in xb0.items()}, train=False, t=t0_, noises=nz0)]
            osec = time.perf_counter() - c0
            del om
            mo = dic.DistilBertModel(E, E, config=dict(n_layers=args.layers, dropout=0.0, attention_dropout=0.0), dtype=args.dtype, device=dev, seed=0)
            mo.load_state(st0)
            mo.eval()
            with torch.no_grad():
                og = [float(v) for v in dic.train_func(mo, None, {k: torch.from_numpy(v).to(dev) for k, v in xb0.items()}, train=False, t=t0_, noises=nz0)]
            oracle_rel = {k: round(abs(a_ - b_) / abs(b_), 8) for k, a_, b_ in zip(("total", "x_t", "x_1", "prob"), og, oref)}
            oracle_rel.update({"mode": args.dtype, "tolerance": 1e-4, "oracle_losses": [round(v, 5) for v in oref], "oracle_seconds": round(osec, 1),
                               "what": f"eval step (dropout off) of the benchmarked engine at the bench shape (B={B}, S={S}, {args.layers} layers, synthetic weights) against "
                                       "oracle/ref_model.py on the host cores, same batch / timesteps / noise"})
            del mo
            torch.cuda.empty_cache()
        except Exception as e:                    # an extra leg never takes the headline line down with it
            leg_errors['loss_rel_vs_cpu_oracle'] = f"{type(e).__name__}: {e}"[:400]
    if extras and (L, w, S, args.layers) == (16, 0.0, 1, 12):
        try:
            dic.cfg.update(BATCH_SIZE=8, SAMPLE_SIZE=100)
            mr = dic.DistilBertModel(E, E, config=dict(n_layers=6, dropout=0.1, attention_dropout=0.1), dtype=args.dtype, device=dev, seed=0)
            trr = dic.AdamW(mr.parameters(), lr=1e-4)
            xr = {k: torch.from_numpy(v).to(dev) for k, v in dic.synth.batch(8, L, 30522, seed=1).items()}
            for _ in range(3):
                dic.train_func(mr, trr, xr)
            torch.cuda.synchronize()
            c0 = time.perf_counter()
            nr = 10
            for _ in range(nr):
                orr = dic.train_func(mr, trr, xr)
            torch.cuda.synchronize()
            dr = (time.perf_counter() - c0) / nr
            ref_faithful = {"value": round(8 / dr, 1), "unit": "captions/s", "ms_per_step": round(dr * 1e3, 3), "steps": nr, "loss": round(float(orr[0]), 4),
                            "sequences_per_step": "100 x 8 x_t + 8 x_1", "dtype": args.dtype,
                            "what": "the reference's own defaults (CLIP-DDPM.py:55-119: SAMPLE_SIZE 100, batch 8, DistilBertConfig() = 6 layers); cpu_baseline_config1 "
                                    "is the CPU port at this shape"}
            del mr, trr
            torch.cuda.empty_cache()
        except Exception as e:
            leg_errors['reference_faithful'] = f"{type(e).__name__}: {e}"[:400]
        finally:
            configure(L, w)

    # ---- CPU baseline: the oracle (a port of the reference step) on this host's cores, bounded sample, rank 0 at N=1 only
    cpu = cpu1 = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.quick:
        try:
            cpu = cpu_leg(dic, torch, E, 16, S, L, args.layers)
            cpu1 = cpu_leg(dic, torch, E, 8, 100, 16, 6)          # BASELINE.json configs[0] / BASELINE.md section 3: B=8, S=100, 6 layers
        except Exception as e:                    # an extra leg never takes the headline line down with it
            leg_errors['cpu_baseline'] = f"{type(e).__name__}: {e}"[:400]

    if rank == 0:
        gf = gflop_per_seq(L, args.layers) * (S + 1)
        guided = f", classifier-free guidance w={w} p=0.2" if w > 0 else ""
        line = {
            "metric": "training captions/sec (seq16, bert-base)" if L == 16 else f"training captions/sec (seq{L}, bert-base)", "value": round(value, 2),
            "unit": "captions/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.dtype in ("bf16r", "bf16w") else args.dtype, "data": "synthetic",
            "config": {"workload": f"train_func: B={B}/GPU x S={S} (+x_1 pass) = {(S + 1) * B} sequences x {L}+2 tokens (an unguided text row is "
                                   f"skipped), {args.layers}-layer DistilBERT-width denoiser, concat fusion, linear beta T=100, dropout 0.1, AdamW{guided}",
                       "precision_mode": args.dtype, "global_batch": world * B, "seq_len": L, "sample_size": S, "n_layers": args.layers,
                       "parallelism": f"dp{world}", "loss": round(loss_val, 4), "launch": graph_note,
                       "algorithmic_tflop_per_s": round(value * gf / 1e3, 2),
                       "executed_tflop_per_s": round(value * gflop_per_seq(L, args.layers, L + 1 if w <= 0 else L + 2) * (S + 1) / 1e3, 2)},
            "roofline": roof, "cpu_baseline": cpu, "cpu_baseline_config1": cpu1, "sustained": sustained, "bf16_vs_fp32_loss_rel": dtype_delta,
            "parity_fast_mode": parity_fast, "throughput_mode": throughput_mode, "fp32_mode": fp32_mode,
            "sampling": sampling, "seq32_cfg": seq32, "data_parallel": dp_info,
            "loss_rel_vs_cpu_oracle": oracle_rel, "reference_faithful_S100_B8_6layer": ref_faithful,
            # every switch that differs from the shipped configuration (diffusion-image-captioning_amd/options.py); {} = the configuration the tests assert
            "options_non_default": OPT.non_default(),
        }
        if leg_errors:
            line["leg_errors"] = leg_errors      # an extra leg that raised: its object above is null, the headline value is unaffected
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

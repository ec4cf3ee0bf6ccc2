"""Data parallelism over the 8 GPUs of one node: one process per GPU, ONE RCCL all-reduce per step.

The reference is single-device (no torch.distributed anywhere).  The path shards naturally: batch items are
independent and every loss term is a mean over the batch dimension (ref :78, :436-437), so summing the per-rank
gradients and dividing by world_size is exact.  All 108 gradient tensors live in one flat fp32 buffer
(`ParamStore.G`), so the exchange is a single `all_reduce` of that buffer -- on MI355X's fully connected xGMI that is
one direct reduce-scatter + all-gather across all 7 links, not 108 latency-bound small collectives.  The 1/world
factor is folded into the AdamW kernel (`grad_scale`), not a separate pass over the buffer.

Reference caveats kept:
  * one t-vector per step is shared by the whole *global* batch (ref :461) -> every rank draws it from an identically
    seeded generator (no per-step collective);
  * the noise eps, the dropout masks and the guidance uniforms are per item -> `configure_model_for_rank` mixes the rank into
    all three seeds;
  * CFG forces rows 0/1 of the batch to unguided/guided (ref :408-409) -> only rank 0 does.
Backend "nccl" IS RCCL on ROCm; the CPU tests run the same code with "gloo".
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def is_initialized() -> bool:
    return dist.is_available() and dist.is_initialized()


def world_size() -> int:
    return dist.get_world_size() if is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if is_initialized() else 0


def init_from_env(backend: str | None = None):
    """Initialise from torchrun's RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* environment; no-op for a single process."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws <= 1 or is_initialized():
        return rank(), world_size(), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("DIC_DIST_SHARE_GPU", "0") == "1":
        local = 0                  # test rig: every rank on GPU 0 (with DIC_DIST_BACKEND=gloo -- RCCL refuses two ranks on one device)
    if backend is None:
        backend = os.environ.get("DIC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(local)
    from datetime import timedelta
    from .options import OPT
    r_ = int(os.environ["RANK"])
    try:
        # a finite timeout for the rendezvous AND for every later collective: a peer that never arrives / dies mid-step must end the job with a
        # message, not leave the survivors in a collective forever (options.dp_timeout_s)
        dist.init_process_group(backend=backend, rank=r_, world_size=ws, timeout=timedelta(seconds=int(OPT.dp_timeout_s)))
    except Exception as e:
        raise RuntimeError(f"data-parallel start-up failed on rank {r_} of {ws} (backend {backend}, MASTER_ADDR={os.environ.get('MASTER_ADDR')}, "
                           f"MASTER_PORT={os.environ.get('MASTER_PORT')}, timeout {OPT.dp_timeout_s} s): {type(e).__name__}: {e} -- every rank must be "
                           "started with the same WORLD_SIZE / MASTER_* and reach init_from_env; on ROCm keep HSA_ENABLE_IPC_MODE_LEGACY=0") from e
    return dist.get_rank(), ws, local


def shard(batch: dict, rank_: int | None = None, world: int | None = None) -> dict:
    """Even split of a global batch dict along dim 0 (global B must be divisible by world_size)."""
    r = rank() if rank_ is None else rank_
    w = world_size() if world is None else world
    out = {}
    for k, v in batch.items():
        n = len(v)
        assert n % w == 0, f"global batch {n} not divisible by world size {w}"
        out[k] = v[r * (n // w):(r + 1) * (n // w)]
    return out


def allreduce_flat(flat: torch.Tensor) -> torch.Tensor:
    """Sum `flat` across ranks in place with ONE collective."""
    if is_initialized() and world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def exchange_group(i: int, n_layers: int, group: int):
    """Layers finish in descending order; when layer i finishes, which layers [first, last) are exchanged now (or None)?
    Groups [.., 2g..3g), [g..2g), then the last g layers one by one: what is exchanged AFTER the backward has nothing left to hide
    it, so the final pieces are kept small."""
    if i >= group and i % group != 0:
        return None
    return (i, min(i + group, n_layers)) if i >= group else (i, i + 1)


class GradReducer:
    """Overlaps the gradient exchange with the backward pass: the flat buffer is laid out layer by layer, so as soon as the
    backward of a group of encoder layers has written its slice, that slice (3 layers = 85 MB fp32) is all-reduced asynchronously
    on RCCL's stream while the layers below are still computing; only the small tail (embeddings, MLM-head transform, CLIP
    projections) is reduced after the backward.  Still one logical exchange of `G` per step -- just issued in a few pieces."""

    def __init__(self, model):
        self.G = model.params.G
        self.store = model.params
        self.handles = []            # (work handle, lo, hi) in issue order
        # options.force_reducer: run the exchange path at world size 1 too (single-GPU test of the data-parallel code path)
        from .options import OPT
        self.active = is_initialized() and (world_size() > 1 or OPT.force_reducer)
        self.group = max(1, int(OPT.dp_group))
        # options.dp_single: north_star's literal design -- exactly ONE all-reduce of the whole flat buffer, after the backward (nothing
        # overlaps it; A/B partner of the sliced default on the first multi-GPU run)
        self.single = bool(OPT.dp_single)
        # options.dp_cu_cap = n: while a slice is on the wire, the backward's persistent GEMMs keep to n CUs' worth of workgroups so that RCCL's
        # kernels find free CUs (the 256-column GEMM holds 128 KB of LDS on every CU it runs on); 0 = no cap
        self.cu_cap = int(OPT.dp_cu_cap)
        self.model = model
        self.n_collectives = 0
        self.timing = bool(OPT.dp_timing) and torch.cuda.is_available()
        self._ev = []

    def layer_done(self, i):
        """Called by Denoiser.backward right after layer i's parameter gradients are complete (layers finish in descending
        order).  Layers are exchanged in groups of options.dp_group (default 3: 85 MB per collective at 12 layers -- xGMI rings are
        per-link bound, fewer and larger collectives use them better than one per layer) as soon as a group is complete."""
        if not self.active or self.single:
            return
        rng = exchange_group(i, self.store.n_layers, self.group)
        if rng is None:
            return
        last = rng[1]
        lo = self.store.off(f"L{i}.Wqkv")
        hi = self.store.off(f"L{last}.Wqkv") if last < self.store.n_layers else self.store.off("pos")
        self._issue(lo, hi)
        ops = getattr(self.model, "ops", None)
        if self.cu_cap > 0 and ops is not None:
            ops.default_cu_cap = self.cu_cap

    def _issue(self, lo, hi):
        ev0 = None
        if self.timing:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record()
        try:
            work = dist.all_reduce(self.G[lo:hi], op=dist.ReduceOp.SUM, async_op=True)
        except Exception as e:
            self.handles = []
            raise RuntimeError(f"data-parallel gradient exchange could not be issued on rank {rank()} of {world_size()}: {type(e).__name__}: {str(e)[:300]} "
                               "-- a peer rank has exited or the communicator is broken") from e
        self.handles.append((work, lo, hi, ev0))
        self.n_collectives += 1

    def finish(self, trainer=None):
        """Reduce the tail, then hand every slice to the optimizer as soon as ITS exchange has finished: with the fused AdamW the
        update of the early (deep) layers runs while the last slices are still on the wire; `trainer.step()` afterwards only
        covers what is left.  Other optimizers get the plain wait-all + 1/world scaling."""
        if not self.active:
            return
        tail = self.store.off("pos")
        ops = getattr(self.model, "ops", None)
        if ops is not None:
            ops.default_cu_cap = 0
        if self.single:
            self._issue(0, self.store.numel)
        else:
            self._issue(tail, self.store.numel)
        w = world_size()
        streamed = trainer is not None and hasattr(trainer, "step_range") and hasattr(trainer, "begin_step")
        if streamed:
            trainer.grad_scale = 1.0 / w
            trainer.begin_step()
        for h, lo, hi, ev0 in self.handles:
            try:
                h.wait()                               # the current stream waits for this slice only
            except Exception as e:                     # a peer died / the collective timed out: one clear error instead of a stack of backend noise
                self.handles = []
                raise RuntimeError(f"data-parallel gradient exchange failed on rank {rank()} of {w} (elements [{lo}, {hi}) of the flat gradient buffer): "
                                   f"{type(e).__name__}: {str(e)[:300]} -- a peer rank has exited or stalled; the step was NOT applied on this rank") from e
            if ev0 is not None:
                ev1 = torch.cuda.Event(enable_timing=True)
                ev1.record()
                self._ev.append((ev0, ev1))
            if streamed and hi <= tail and not self.single:
                trainer.step_range(lo, hi)
        self.handles = []
        GradReducer.last = self
        if not streamed:
            if trainer is not None and hasattr(trainer, "grad_scale"):
                trainer.grad_scale = 1.0 / w
            else:
                self.G.mul_(1.0 / w)


    def allreduce_ms(self):
        """Issue-to-completion time of this step's collectives on the compute stream (options.dp_timing), summed; synchronises."""
        if not self._ev:
            return None
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self._ev)


GradReducer.last = None


def allreduce_grads(model, trainer=None):
    """One all-reduce of the model's flat gradient buffer; the mean is taken by the optimizer's grad_scale."""
    w = world_size()
    if w <= 1:
        return
    allreduce_flat(model.params.G)
    if trainer is not None and hasattr(trainer, "grad_scale"):
        trainer.grad_scale = 1.0 / w
    else:
        model.params.G.mul_(1.0 / w)


def allreduce_scalars(*vals):
    """Mean of a few logging scalars across ranks (one tiny collective)."""
    if world_size() <= 1:
        return vals
    t = torch.stack([v.detach().float().reshape(()) for v in vals])
    dist.all_reduce(t)
    t /= world_size()
    return tuple(t[i] for i in range(len(vals)))


_RANK_MIX = 0x9E3779B97F4A7C15      # odd 64-bit constant: rank r shifts every per-rank seed by r * _RANK_MIX (mod 2^63)


def rank_seed(base: int, rank_: int | None = None) -> int:
    """Per-rank variant of a seed: identical on rank 0, distinct on every other rank (kept below 2^63: the C-ABI takes uint64,
    torch generators take int64)."""
    r = rank() if rank_ is None else rank_
    return (int(base) + r * _RANK_MIX) & 0x7FFFFFFFFFFFFFFF


def configure_model_for_rank(model, rank_: int | None = None):
    """What differs between the ranks of a data-parallel job: the noise eps, the dropout masks and the
    guidance uniforms are per item, so every rank draws its own (seeds mixed with the rank); the t-vector stays shared
    (`share_timestep_seed`: rank 0's counter); the forced unguided/guided rows 0/1 of ref :408-409 exist once per GLOBAL batch, i.e. on rank 0."""
    from . import diffusion
    r = rank() if rank_ is None else rank_
    model.rank_rows_forced = r == 0
    model.set_dropout_seed(rank_seed(model.dropout_seed_base, r))
    diffusion.seed_noise(rank_seed(diffusion._state.get("noise_base", diffusion.NOISE_SEED_BASE), r))
    gbase = diffusion._state.get("guidance_base", diffusion.GUIDANCE_SEED_BASE)
    diffusion.seed_guidance(rank_seed(gbase, r), model.device)
    share_timestep_seed()
    return model


def share_timestep_seed():
    """Every rank continues the timestep stream from rank 0's counter: one small broadcast at an explicit synchronisation point (model
    configuration, the start of harness.fit, after a checkpoint restore) -- every rank must call it; none per step."""
    from . import diffusion
    if "t_seed" not in diffusion._state:
        diffusion._state["t_seed"] = diffusion._default_t_seed()     # materialise the default (torch.initial_seed()-derived) start; no collective in there
    if is_initialized() and world_size() > 1:
        box = [diffusion._state["t_seed"]]
        dist.broadcast_object_list(box, src=0)
        diffusion._state["t_seed"] = int(box[0])
    diffusion._state["t_seed_shared"] = True
    return diffusion._state["t_seed"]


def assert_shared_timestep_seed():
    """The ranks must have drawn the same number of t-vectors (train AND validation steps): a rank-0-only validation, or uneven validation
    shards, would silently desynchronise t.  One all-gather of an int; harness.fit calls it once per epoch."""
    from . import diffusion
    if not (is_initialized() and world_size() > 1):
        return
    seeds = [None] * world_size()
    dist.all_gather_object(seeds, diffusion._state.get("t_seed"))
    if len(set(seeds)) != 1:
        raise RuntimeError(f"data-parallel ranks disagree on the timestep stream (t_seed per rank: {seeds}): every rank must call "
                           "train_func / validate the same number of times")

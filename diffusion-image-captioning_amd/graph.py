"""The training step as ONE hipGraph: `train_func(model, trainer, x)` (ref CLIP-DDPM.py:458-486) is captured once -- both streams, every
event between them, the per-layer AdamW launches -- and replayed; a replay is a single host call instead of ~330 launches, the ~0.3 ms of
launch gaps in the forward disappear and the order of the two streams' kernels is the same on every step.

What a capture would freeze but a step must vary is read from device memory instead (include/dic_hip.h, `dic_step_ctx_set`): a device step
counter, bumped by the graph's first node, shifts every dropout / noise / timestep seed by exactly what the host adds between two eager
steps, selects AdamW's bias corrections from a table of host-computed factors, and picks the result slot the step's losses are written
to.  A graphed run is therefore BIT-IDENTICAL to the eager run with the same seeds (tests/test_gpu_e2e.py), and the host-side counters are
advanced alongside.  Eager and graphed steps can be mixed in either order: every call compares the host counters (dropout / noise / timestep
seeds, optimizer step, result slot) with what the last replay left behind, and anything that moved them in between -- an eager `train_func`,
a `validate`, a restored checkpoint -- makes the call RE-CAPTURE from the current counters instead of replaying stale seeds.  The captured
graph holds raw pointers into the encoder / rounding workspaces: both are pinned against the model's workspace eviction for as long as this
object lives (`release()` unpins them).

Scope: one GPU, fused `AdamW`, no classifier-free guidance (its stacked batch changes size every step), fixed batch shape.  The learning rate
is a kernel argument: changing `param_groups[0]["lr"]` (per epoch in the reference, ref :520-522) re-captures.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .config import cfg
from . import diffusion


class GraphedTrainStep:
    """step = GraphedTrainStep(model, trainer, x);  l, x_t_loss, x_1_loss, prob_loss = step()  or  step(next_batch)."""

    def __init__(self, model, trainer, x, warmup: int = 2):
        from . import parallel
        if float(cfg.CLASSIFIER_FREE_WEIGHT) > 0:
            raise ValueError("GraphedTrainStep: classifier-free guidance changes the stacked batch's size every step -- not capturable")
        if not isinstance(trainer, diffusion.AdamW):
            raise TypeError("GraphedTrainStep needs the fused dic.AdamW (its bias corrections come from a device table under replay)")
        if parallel.world_size() > 1 or model.te:
            raise ValueError("GraphedTrainStep: single-GPU main path only")
        self.model, self.trainer = model, trainer
        dev = model.device
        self.x = {k: (v.to(dev).contiguous().clone() if torch.is_tensor(v) else v) for k, v in x.items()}
        self.ctr = torch.zeros(1, dtype=torch.int64, device=dev)
        self.n = 0                         # host mirror of the device counter
        self.replays = 0
        self.captures = 0
        self._pinned, self._pin_box, self._finalizer = (), [], None      # workspaces this object keeps out of the engine's eviction (see _capture)
        for _ in range(warmup):            # allocates workspaces / the side stream, loads every kernel: nothing may allocate while capturing
            diffusion.train_func(model, trainer, self.x)
        torch.cuda.synchronize()
        self._capture()

    # one capture is replayed for at most HORIZON steps: result slots and the AdamW table are sized for it.  Half the ring, so that a
    # capture can always continue from the current result slot or wrap to slot 0 without touching a slot younger than LOSS_RING / 2 calls
    HORIZON = diffusion.LOSS_RING // 2 - 4

    def _capture(self):
        model, tr = self.model, self.trainer
        L = _lib.lib()
        dev = model.device
        g = tr.param_groups[0]
        b1, b2 = g["betas"]
        self.lr = float(g["lr"])
        # AdamW's bias corrections for the captured step (optimizer step t0 + 1) and the HORIZON after it, with the host's own arithmetic:
        # Python doubles rounded to fp32, then 1/sqrt in fp32 exactly as dic_adamw does (misc.hip)
        ts = np.arange(tr.t + 1, tr.t + 2 + self.HORIZON, dtype=np.float64)
        bc1 = np.array([np.float32(1.0 - b1 ** int(t)) for t in ts], dtype=np.float32)
        bc2 = np.array([np.float32(1.0 - b2 ** int(t)) for t in ts], dtype=np.float32)
        tab = np.stack([bc1, np.float32(1.0) / np.sqrt(bc2)], 1).astype(np.float32)
        self.table = torch.from_numpy(tab).to(dev).contiguous()
        # the loss kernels of replay k write result slot slot0 + k of the workspace's ring
        B, L_ = self.x["input_ids"].shape
        ws = model._workspace((cfg.SAMPLE_SIZE + 1) * B, L_, model.concat and cfg.DROP_UNUSED_TEXT_ROW)
        sc = ws["loss_sc"]
        if sc["slot"] + self.HORIZON + 3 >= diffusion.LOSS_RING:
            sc["slot"] = 0                  # replays write slots slot0 .. slot0 + HORIZON without wrapping: restart at the ring's (oldest) head
        self.sc = sc
        # the graph bakes in the addresses of these workspaces: keep them out of Denoiser._evict's reach -- until this object goes away or
        # re-captures onto another pair (pins are counted: several graphed steps may share a workspace)
        cw = model._ce_workspace((cfg.SAMPLE_SIZE + 1) * B * L_)
        self._unpin()
        for w_ in (ws, cw):
            w_["pinned"] = w_.get("pinned", 0) + 1
        self._pinned = (ws, cw)
        if self._finalizer is None:
            import weakref
            self._finalizer = weakref.finalize(self, GraphedTrainStep._unpin_list, self._pin_box)
        self._pin_box[:] = [ws, cw]
        self.stride_noise = (2 if cfg.X_0_PREDICTION else 3) * 0x9E3779B1          # diffusion._next_seed per q_sample call
        torch.cuda.synchronize()
        _lib.check(L.dic_step_ctx_set(self.ctr.data_ptr(), self.n + 1, self.stride_noise, self.table.data_ptr()), "step_ctx_set")
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                _lib.check(L.dic_step_advance(self.ctr.data_ptr(), torch.cuda.current_stream().cuda_stream), "step_advance")
                diffusion.train_func(model, tr, self.x)          # (this advanced the host-side seeds / step counters by one step: replay 1 IS that step)
        finally:
            L.dic_step_ctx_set(0, 0, 0, 0)
        self.graph = graph
        self.slot0 = sc["slot"]
        self.k = 0                          # replays since this capture
        self.captures += 1
        self._expect = self._host_state()   # (the capture advanced the host counters by one step: replay 1 IS that step)

    def _host_state(self):
        return (int(self.model._seed), int(diffusion._state["noise_seed"]), int(diffusion._state.get("t_seed", 0)), int(self.trainer.t),
                int(self.sc["slot"]), bool(self.model.training))

    @staticmethod
    def _unpin_list(box):
        for w in box:
            n = int(w.get("pinned", 0)) - 1
            if n > 0:
                w["pinned"] = n
            else:
                w.pop("pinned", None)
        box[:] = []

    def _unpin(self):
        GraphedTrainStep._unpin_list(self._pin_box)
        self._pinned = ()

    def release(self):
        """Unpin the workspaces (the graph must not be replayed afterwards).  Dropping the object does the same (weakref finaliser)."""
        self._unpin()
        self.graph = None

    def _advance_host(self):
        """What one eager train_func call adds on the host, so that eager steps can follow graphed ones (and the checkpointed rng state is right)."""
        self.model._seed += 64
        diffusion._state["noise_seed"] += self.stride_noise
        diffusion._state["t_seed"] = diffusion._state.get("t_seed", 0) + 1
        self.trainer.t += 1
        self.sc["slot"] += 1

    def __call__(self, x=None):
        if x is not None and x is not self.x:
            for k, v in x.items():
                if torch.is_tensor(v):
                    self.x[k].copy_(v, non_blocking=True)
        if self.graph is None:
            raise RuntimeError("GraphedTrainStep: called after release()")
        # anything that moved the host counters since the last replay (an eager step, validate(), a restored checkpoint) consumed seeds /
        # optimizer steps / result slots the device counter knows nothing about: re-capture from the counters as they are now
        if self.k >= self.HORIZON or float(self.trainer.param_groups[0]["lr"]) != self.lr or self._host_state() != self._expect:
            self._capture()
        if self.k > 0:
            self._advance_host()
        self.graph.replay()
        self.k += 1
        self.n += 1
        self.replays += 1
        self._expect = self._host_state()
        o = self.sc["ring"][self.slot0 + self.k - 1]
        return o[7], o[0], o[1], o[6]

#!/usr/bin/env python3
"""RCCL smoke for the data-parallel pieces on however many ranks torchrun gives (1 on the single-GPU box):
init from env, broadcast of the shared t-vector, ONE all-reduce of a flat gradient-sized buffer, scalar all-reduce."""
import importlib
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dic = importlib.import_module("diffusion-image-captioning_amd")

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dist.init_process_group("nccl", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
dev = torch.device("cuda", local)
dic.cfg.update(STEP_TOT=100)
dic.parallel.share_timestep_seed()                         # every rank continues rank 0's timestep counter ...
t = importlib.import_module("diffusion-image-captioning_amd.diffusion")._draw_t(4, dev)
tt = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
dist.all_gather(tt, t)                                      # ... so the t-vector drawn on the device is the same everywhere, with no per-step collective
assert all(torch.equal(tt[0], x) for x in tt), "ranks drew different timesteps"
g = torch.ones(86_830_848 + 4096, device=dev)             # 12-layer flat gradient buffer
dist.all_reduce(g)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    dic.parallel.allreduce_flat(g) if dist.get_world_size() > 1 else dist.all_reduce(g)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
(l,) = dic.parallel.allreduce_scalars(torch.tensor(3.0, device=dev))
dist.barrier()
if dist.get_rank() == 0:
    print(f"rccl ok: world={dist.get_world_size()} t={t.flatten().tolist()} allreduce({g.numel()*4/1e6:.0f} MB) {dt*1e3:.2f} ms scalar={float(l)}")
dist.destroy_process_group()

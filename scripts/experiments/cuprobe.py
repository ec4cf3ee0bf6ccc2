#!/usr/bin/env python3
"""Streaming bandwidth (2 reads + 1 write of 27 MB each) against the number of 512-thread workgroups, bytes per lane and accesses in flight.
Build on the GPU box: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o /tmp/libcuprobe.so scripts/experiments/cuprobe.hip"""
import ctypes as C, torch
L = C.CDLL("/tmp/libcuprobe.so")
L.cuprobe.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p]
n = 17408 * 768 * 2
a, b, c = (torch.zeros(n, dtype=torch.uint8, device="cuda") for _ in range(3))
st = torch.cuda.current_stream().cuda_stream
for vb in (8, 16):
    for u in (1, 4, 8):
        line = f"{vb:2d} B/lane, {u} in flight:"
        for grid in (32, 64, 128, 256, 512, 1024):
            for _ in range(3):
                assert L.cuprobe(a.data_ptr(), b.data_ptr(), c.data_ptr(), n, grid, vb, u, st) == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                L.cuprobe(a.data_ptr(), b.data_ptr(), c.data_ptr(), n, grid, vb, u, st)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 20
            line += f"   {grid:4d} wg {us:6.1f} us {3 * n / us / 1e6:5.2f} TB/s"
        print(line, flush=True)

#!/usr/bin/env python3
import ctypes as C, os, torch
L = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ab", "libwprobe.so"))
L.wprobe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_longlong, C.c_void_p]
def t(grid, nth, bpw, iters=30):
    buf = torch.empty(grid * bpw, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3): L.wprobe(buf.data_ptr(), grid, nth, bpw, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): L.wprobe(buf.data_ptr(), grid, nth, bpw, st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
print("total 28 MB written by G workgroups of T threads (contiguous per workgroup):")
tot = 28 * 1024 * 1024
for grid, nth in ((216, 512), (256, 512), (256, 1024), (512, 256), (512, 512), (1024, 256), (2048, 256), (4096, 256), (8192, 256), (28672, 64)):
    bpw = tot // grid // 4096 * 4096
    us = t(grid, nth, bpw)
    print(f"  grid {grid:6d} x {nth:4d} thr, {bpw//1024:5d} KB each: {us:6.1f} us  {grid*bpw/us/1e6:6.2f} TB/s")
print("total 113 MB:")
tot = 113 * 1024 * 1024
for grid, nth in ((256, 512), (512, 256), (864, 512), (2048, 256), (8192, 256)):
    bpw = tot // grid // 4096 * 4096
    us = t(grid, nth, bpw)
    print(f"  grid {grid:6d} x {nth:4d} thr, {bpw//1024:5d} KB each: {us:6.1f} us  {grid*bpw/us/1e6:6.2f} TB/s")

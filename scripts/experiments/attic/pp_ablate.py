#!/usr/bin/env python3
"""Times the plain forward GEMM (k-contiguous operands, bf16 out, no bias) of every library build under ab/pp_*/ on a few shapes, cold operands,
each build in its own process (DIC_HIP_LIB).  Run on the GPU box after scripts/experiments/pp_ablate.sh."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import ctypes as C, importlib, os, sys, torch
sys.path.insert(0, %r)
dic = importlib.import_module("diffusion-image-captioning_amd")
L = dic.lib(); GP = dic._lib.GemmParams; bf = torch.bfloat16
L.dic_gemm_set_variant(int(os.environ.get("PPV", "1")))
def run(M, N, K, iters=20):
    per = (M * K + N * K + M * N) * 2
    nset = max(2, min(24, int(1.5e9 // per)))
    sets = []
    for _ in range(nset):
        A = torch.randn(M, K, device="cuda").to(bf); B = torch.randn(N, K, device="cuda").to(bf); Cc = torch.empty(M, N, device="cuda", dtype=bf)
        sets.append((GP(A=A.data_ptr(), B=B.data_ptr(), C=Cc.data_ptr(), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, tile=256), A, B, Cc))
    st = torch.cuda.current_stream().cuda_stream
    for g, *_ in sets: assert L.dic_gemm(1, 0, 0, 0, C.byref(g), st) == 0
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = max(iters, nset); e0.record()
        for i in range(n): L.dic_gemm(1, 0, 0, 0, C.byref(sets[i %% nset][0]), st)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return 2.0 * M * N * K / best / 1e9
print(os.environ.get("TAG"), " ".join(f"{run(*s):7.0f}" for s in [(17408, 2304, 768), (17408, 768, 768), (17408, 768, 3072), (4096, 4096, 4096), (8192, 8192, 8192)]), flush=True)
''' % ROOT
print("variant                qkv   outproj   ffn2    4096^3  8192^3   (TFLOP/s, cold operands, best of 3)")
libs = [("shipped pp", None, "1"), ("shipped lockstep", None, "0")] + [(os.path.basename(d), os.path.join(d, "libdic_hip.so"), "1") for d in sorted(glob.glob(os.path.join(ROOT, "ab", "pp_*")))]
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    for tag, lib, ppv in libs:
        env = dict(os.environ, TAG=f"{tag:20s}", PPV=ppv)
        if lib: env["DIC_HIP_LIB"] = lib
        subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)

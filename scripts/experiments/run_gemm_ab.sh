#!/bin/bash
# GEMM tests + microbenchmark of the current build vs A/B builds under ab/ + phase trace; output under gpurun_out/g1/
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g1
true
true
run() { # name lib env...
  name=$1; lib=$2; shift 2
  if [ -n "$lib" ]; then export DIC_HIP_LIB=$GRAFT_REPO_ROOT/ab/$lib; else unset DIC_HIP_LIB; fi
  env "$@" COLD=${COLD:-0} TILE=256 timeout 300 python scripts/gemm_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/g1/bench_${name}.txt
  unset DIC_HIP_LIB
}
NAMES="r1 new pf2 pf4 im1 i3 prio"
run r1 libdic_r1.so A=1
run new "" A=1
for v in im1 i1 plain; do run $v libdic_$v.so A=1; done
python - <<'PY'
import re
names="r1 new im1 i1 plain".split()
rows={}
for n in names:
    for line in open(f"gpurun_out/g1/bench_{n}.txt"):
        m=re.match(r"(.{34}) M=\s*(\d+) N=\s*(\d+) K=\s*(\d+) split=\s*(\d+)\s+([\d.]+) us\s+([\d.]+) TFLOP/s",line)
        if m: rows.setdefault((m.group(1).strip(),m.group(5)),{})[n]=float(m.group(6))
print(f"{'':36s}"+"".join(f"{n:>9s}" for n in names))
for (k,sp),v in rows.items():
    print(f"{k[:30]:30s} s{sp:>2s}  "+"".join(f"{v.get(n,0):9.1f}" for n in names))
PY
DIC_HIP_LIB=$GRAFT_REPO_ROOT/ab/libdic_trace.so python scripts/experiments/gemm_trace.py 2>&1 | grep -v amdgpu.ids > gpurun_out/g1/trace.txt
grep -E "==|tile|K-steps" gpurun_out/g1/trace.txt | cut -c1-200 | head -24

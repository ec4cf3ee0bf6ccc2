#!/bin/bash
# GEMM microbenchmark of the current build vs A/B builds under ab/ (usage: run_gemm_ab.sh "<ab lib names>"; COLD=1 rotates operands out of the caches)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g1
NAMES="new $1"
run() { name=$1; lib=$2
  if [ -n "$lib" ]; then export DIC_HIP_LIB=$GRAFT_REPO_ROOT/ab/$lib; else unset DIC_HIP_LIB; fi
  COLD=${COLD:-0} TILE=256 timeout 300 python scripts/gemm_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/g1/bench_${name}.txt
  unset DIC_HIP_LIB
}
run new ""
for v in $1; do run $v libdic_$v.so; done
NAMES="$NAMES" python - <<'PY'
import re, os
names=os.environ["NAMES"].split()
rows={}
for n in names:
    for line in open(f"gpurun_out/g1/bench_{n}.txt"):
        m=re.match(r"(.{34}) M=\s*(\d+) N=\s*(\d+) K=\s*(\d+) split=\s*(\d+)\s+([\d.]+) us\s+([\d.]+) TFLOP/s",line)
        if m: rows.setdefault((m.group(1).strip(),m.group(5)),{})[n]=float(m.group(6))
print(f"{'':36s}"+"".join(f"{n:>9s}" for n in names))
for (k,sp),v in rows.items():
    print(f"{k[:30]:30s} s{sp:>2s}  "+"".join(f"{v.get(n,0):9.1f}" for n in names))
PY

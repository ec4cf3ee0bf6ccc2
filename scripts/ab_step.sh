#!/bin/bash
# In-step A/B of environment switches on ONE box: every variant (a quoted "VAR=.. VAR=.." string, "-" = defaults) is run REPS times, interleaved,
# as `bench.py --quick`; prints captions/s, ms/step and the roofline leg's single-stream GEMM time per step for each run.
#   scripts/ab_step.sh [-r REPS] [-s STEPS] [-x "extra bench args"] "DIC_GELU_D=0" "DIC_GELU_D=1" ...
REPS=2; STEPS=40; EXTRA=""
while getopts "r:s:x:" o; do case $o in r) REPS=$OPTARG;; s) STEPS=$OPTARG;; x) EXTRA=$OPTARG;; esac; done
shift $((OPTIND-1))
cd "$(dirname "$0")/.."
for r in $(seq $REPS); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then envs=""; else envs="$v"; fi
    line=$(env $envs python bench.py --quick --steps $STEPS $EXTRA 2>/dev/null | tail -1)
    python - "$v" "$line" <<'PY'
import json, sys
v, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
    r = d.get("roofline") or {}
    print(f"{v:60s} {d['value']:9.1f} captions/s  {d['ms_per_step']:7.3f} ms/step   GEMM single-stream {r.get('gemm_ms_per_step')} ms  frac {r.get('frac')}  launches {r.get('launches_per_step')}", flush=True)
except Exception as e:
    print(f"{v:60s} FAILED: {line[:200]}")
PY
  done
done

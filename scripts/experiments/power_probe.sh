#!/bin/bash
# Is the training step POWER-bound?  Samples rocm-smi (socket power, sclk, temperature) while the step runs in a loop, for the default build and with
# the four-wave asm GEMMs on (DIC_GEMM_W4A=1), plus an idle sample and a pure-GEMM loop.  Round 4: in-step A/B showed every UNCHANGED kernel of the
# step running 4-5 % slower in the process that used the denser asm GEMMs -- a chip-wide clock effect, not a kernel effect.
cd "$(dirname "$0")/../.."
smi() { rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|junction|edge" | tr '\n' ';' | sed 's/  */ /g'; echo; }
echo "== idle"; smi
for v in "-" "DIC_GEMM_W4A=1"; do
  echo "== training step loop, $v"
  if [ "$v" = "-" ]; then envs=""; else envs="$v"; fi
  env $envs python bench.py --quick --no-roofline --steps 1500 --warmup 20 > /tmp/pp_bench.log 2>&1 &
  pid=$!
  sleep 14
  for i in 1 2 3 4 5 6; do smi; sleep 1.5; done
  wait $pid
  tail -1 /tmp/pp_bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ->', d['value'], 'captions/s', d['ms_per_step'], 'ms/step')"
done
for shape in "8192 8192 8192" "17408 2304 768"; do
  for m in 0 1; do
    echo "== GEMM loop, w4a=$m, shape $shape"
    python scripts/experiments/gemm_power_loop.py $m $shape 9 > /tmp/pp_gemm.log 2>&1 &
    pid=$!
    sleep 4
    for i in 1 2 3; do smi; sleep 1.2; done
    wait $pid
    grep sustained /tmp/pp_gemm.log | tail -3
  done
done

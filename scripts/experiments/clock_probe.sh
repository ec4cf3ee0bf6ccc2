#!/bin/bash
# sample GPU clock / power while the training bench runs (is the step clock- or power-limited?)
python bench.py --no-cpu-baseline --no-roofline --steps 2500 --warmup 5 > gpurun_out/clk_bench.log 2>&1 &
BP=$!
sleep 14
for i in $(seq 1 30); do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction|hotspot)" | tr '\n' ' ' | sed 's/GPU\[0\]\s*: //g'
  echo
  sleep 1
done
wait $BP
tail -1 gpurun_out/clk_bench.log | cut -c1-150
echo "--- idle"
sleep 3
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo

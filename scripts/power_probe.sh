#!/bin/bash
# socket power / shader clock while bench.py runs (plan 0 region, then plan 1 region): is FC1 plan 1 clock-limited by the power cap?
set -u
mkdir -p gpurun_out
python bench.py --no-cpu-baseline --no-host-scopes --steps ${STEPS:-400} > gpurun_out/power_bench.json 2> gpurun_out/power_bench.err &
BP=$!
sleep ${DELAY:-12}
: > gpurun_out/power_samples.txt
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Power|sclk|Current Socket|fclk|mclk" | tr '\n' ' ' >> gpurun_out/power_samples.txt
  echo >> gpurun_out/power_samples.txt
  sleep 0.05
done
wait $BP
python scripts/summarize.py gpurun_out/power_bench.json
awk 'NR%4==0' gpurun_out/power_samples.txt | head -60

#!/bin/bash
# generic A/B: for each value of env knob $KNOB in $VALUES run bench on workloads $WLS
set -u
mkdir -p gpurun_out
KNOB=${KNOB:-ETHCNN_FC1_VARIANT}
VALUES=${VALUES:-"0 1 2 3"}
WLS=${WLS:-"c2 c3"}
STEPS=${STEPS:-10}
if [ "${PYTEST:-1}" = "1" ]; then python -m pytest tests -m gpu -x -q 2>&1 | tail -3; fi
for wl in $WLS; do for v in $VALUES; do
  env $KNOB=$v python bench.py --workload $wl --no-cpu-baseline --no-host-scopes --steps $STEPS > gpurun_out/ab_${wl}_$v.json 2>gpurun_out/ab_${wl}_$v.err || tail -3 gpurun_out/ab_${wl}_$v.err
done; done
python scripts/summarize.py "gpurun_out/ab_*.json"

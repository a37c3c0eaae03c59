#!/bin/bash
# One short gpurun call: GPU tests (each test bounded), then bench lines for A/B.
#   SKIP_TESTS=1 skips pytest;  AB="name:ENV=val,ENV2=val name2:..." runs bench once per entry (2 repetitions)
set -u
mkdir -p gpurun_out
if [ -z "${SKIP_TESTS:-}" ]; then
  python -m pytest tests -m gpu -x -q --timeout 400 --durations=8 2>&1 | tail -16
fi
for rep in 1 2; do
  for ent in ${AB:-default:}; do
    name=${ent%%:*}; envs=${ent#*:}
    env $(echo $envs | tr ',' ' ') python bench.py --no-cpu-baseline --no-host-scopes --steps 30 ${BENCH_ARGS:-} > gpurun_out/ab_${name}_$rep.json 2> gpurun_out/ab_${name}_$rep.err || tail -3 gpurun_out/ab_${name}_$rep.err
  done
done
python scripts/summarize.py "gpurun_out/ab_*.json"

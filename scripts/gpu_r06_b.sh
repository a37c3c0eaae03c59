#!/bin/bash
# round 6, second call: the rest of the GPU suite behind the guard test, plan-3 overlap / power probe, cold start with parallel peers
set -u
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -25 | tee gpurun_out/r06_gpu_tests_b.txt
python scripts/plan3_power.py 3 2>&1 | tail -20
python scripts/cold_start.py 3 2>&1 | tail -90
bash scripts/gpu_fence_ab.sh 2>&1 | tail -70

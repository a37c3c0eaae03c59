#!/bin/bash
# round 6, first call: the GPU suite at the refactored library (split api, pruned launch plans, accuracy guard, sharded entry) + cold start
set -u
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --timeout 900 2>&1 | tail -25 | tee gpurun_out/r06_gpu_tests_a.txt
python __graft_entry__.py smoke 2>&1 | tail -3
python scripts/cold_start.py 3 2>&1 | tail -80

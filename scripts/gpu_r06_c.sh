#!/bin/bash
# round 6, third call: lazy side streams (cold start again), stage-alone power on real features, the hand-off stress, LDP tail with the diagnostic client
set -u
mkdir -p gpurun_out
rm -f gpurun_out/handoff_stress.txt
python -m pytest tests/test_gpu_cli.py tests/test_gpu_switch_points.py tests/test_gpu_parity.py tests/test_gpu_soak.py -m gpu -q --timeout 900 2>&1 | tail -6
python scripts/plan3_power.py 3 2>&1 | tail -12
python scripts/cold_start.py 3 > /dev/null 2>&1; grep -v "^      " gpurun_out/cold_start.txt
LAUNCHES=100000 python scripts/handoff_stress.py 2>&1 | tail -5
python scripts/ldp_tail.py 1500 > gpurun_out/ldp_tail.txt 2>&1; grep -E "handshake p50" gpurun_out/ldp_tail.txt | cut -c1-230

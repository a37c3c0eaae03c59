#!/bin/bash
# round-5 scratch call: tests of what changed + the A/Bs waiting for a box
set -u
mkdir -p gpurun_out
REPO=$PWD
python -m pytest tests/test_gpu_fast_plan.py tests/test_gpu_lstm.py tests/test_gpu_ldp_native.py tests/test_gpu_e2e.py -m gpu -x -q --timeout 600 2>&1 | tail -5
[ -x scripts/ubench/fc1_tile_f16_probe ] || bash scripts/build_probes.sh
./scripts/ubench/fc1_tile_f16_probe > gpurun_out/fc1_tile_f16_probe.txt 2>&1; cat gpurun_out/fc1_tile_f16_probe.txt
python scripts/ldp_tail.py 1500 > gpurun_out/ldp_tail2.txt 2>&1; grep -v "^    POC\|^    daemon" gpurun_out/ldp_tail2.txt | cut -c1-300

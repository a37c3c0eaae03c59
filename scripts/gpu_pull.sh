#!/bin/bash
# the PULL form of the single-launch pass (ethcnn_small.hip): parity, latency A/B
set -u
mkdir -p gpurun_out
REPO=$PWD
EXP=$REPO/hevc-complexity-reduction_amd/lib_exp/libethcnn.so
timeout 900 python -m pytest tests/test_gpu_small.py tests/test_gpu_parity.py tests/test_gpu_lstm.py tests/test_gpu_robustness.py -x -q -m gpu --timeout 300 2>&1 | tail -4
{ python scripts/latency_host.py
  echo "# --- the same with ETHCNN_PULL=0 (experiments build): DMA first, banded above 1024 CTUs (the round's first form)"; ETHCNN_LIB=$EXP ETHCNN_PULL=0 python scripts/latency_host.py; } > gpurun_out/latency_host.txt 2>&1; cat gpurun_out/latency_host.txt
python scripts/latency_ldp.py > gpurun_out/latency_ldp.txt 2>&1; cat gpurun_out/latency_ldp.txt

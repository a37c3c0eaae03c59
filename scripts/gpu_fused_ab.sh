#!/bin/bash
# fused FC1 + heads + gate launch: parity tests, then stage-alone timings of the fused launch, its FC1 part alone and the separate launches
set -u
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fused.py tests/test_gates_golden.py tests/test_lstm_torch_golden.py -m gpu -x -q --timeout 300 2>&1 | tail -6
for rep in 1 2; do
  ETHCNN_FUSED=0 python bench.py --no-cpu-baseline --no-host-scopes --steps 30 > gpurun_out/ab_sep_$rep.json 2> gpurun_out/ab_sep_$rep.err
  ETHCNN_FUSED=1 python bench.py --no-cpu-baseline --no-host-scopes --steps 30 > gpurun_out/ab_fused_$rep.json 2> gpurun_out/ab_fused_$rep.err
  ETHCNN_FUSED=1 ETHCNN_FUSED_EXP=1 python bench.py --no-cpu-baseline --no-host-scopes --steps 30 > gpurun_out/ab_fc1only_$rep.json 2> gpurun_out/ab_fc1only_$rep.err
done
python scripts/summarize.py "gpurun_out/ab_*.json"

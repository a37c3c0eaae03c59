set -u
mkdir -p gpurun_out
LIBS="base default" WLS="c3 c2" STEPS=30 bash scripts/gpu_ab_lib.sh
python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_lstm.py -x -q -m gpu 2>&1 | tail -3
python scripts/latency_ldp.py > gpurun_out/latency_ldp.txt 2>&1; cat gpurun_out/latency_ldp.txt

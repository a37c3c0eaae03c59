set -u
mkdir -p gpurun_out
./scripts/ubench/clock_probe > gpurun_out/clock_probe.txt 2>&1; cat gpurun_out/clock_probe.txt
LIBS="default agprfc1 agprfc1h" WLS="c3" STEPS=30 bash scripts/gpu_ab_lib.sh
ETHCNN_LIB=$PWD/hevc-complexity-reduction_amd/lib_agprfc1h/libethcnn.so python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stages or frames" 2>&1 | tail -2
python scripts/decision_stability.py 50 > gpurun_out/decision_stability.json 2> gpurun_out/decision_stability.err; tail -9 gpurun_out/decision_stability.err
python -m pytest tests/test_gpu_stability.py -x -q -m gpu 2>&1 | tail -2

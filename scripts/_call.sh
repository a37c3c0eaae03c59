set -u
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
LIBS="base default" WLS="c3 c2" STEPS=30 bash scripts/gpu_ab_lib.sh

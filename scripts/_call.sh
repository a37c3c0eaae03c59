set -u
mkdir -p gpurun_out; rm -f gpurun_out/abl_*.json
LIBS="default prio2 prio3" WLS="c3" STEPS=30 bash scripts/gpu_ab_lib.sh

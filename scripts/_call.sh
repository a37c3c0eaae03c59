set -u
bash scripts/gpu_round.sh
python scripts/c4_full.py > gpurun_out/c4_full_size.txt 2>&1; cat gpurun_out/c4_full_size.txt

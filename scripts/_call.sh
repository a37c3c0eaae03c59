set -u
mkdir -p gpurun_out
ROWS=256,510,924,2040,3696,3927,6088,8160 python scripts/fc1_rows.py -1 0 1 2 3 4 5 21 22 23 24 25 26 > gpurun_out/fc1_rows.txt 2>&1; cat gpurun_out/fc1_rows.txt

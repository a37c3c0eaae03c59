set -u
mkdir -p gpurun_out
./scripts/ubench/clock_probe > gpurun_out/clock_probe.txt 2>&1; grep -E "w3" gpurun_out/clock_probe.txt | cut -c1-120

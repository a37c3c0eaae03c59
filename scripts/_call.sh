set -u
bash scripts/gpu_round.sh
WL=c2 SKIP_TESTS=1 bash -c 'cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_c2 -o c2 -- python /root/repo/bench.py --workload c2 --no-cpu-baseline --no-host-scopes > /root/repo/gpurun_out/prof_c2.log 2>&1'
head -8 gpurun_out/prof_c2/c2_kernel_stats.csv | cut -c1-160

set -u
mkdir -p gpurun_out
python bench.py --no-cpu-baseline --steps 5 2>&1 | tail -3 | cut -c1-600
for cpus in 0-63 64-127; do
taskset -c $cpus python bench.py --no-cpu-baseline --steps 5 > gpurun_out/numa_$cpus.json 2>gpurun_out/numa.err || tail -3 gpurun_out/numa.err; python -c "
import json; d=json.load(open('gpurun_out/numa_$cpus.json')); h=d['host_scopes']; print('cpus $cpus', d['value'], h['s2_host_to_host_ctus_per_s'], h['s3_file_to_file_ctus_per_s'])"
done
cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' '; echo

#!/bin/bash
# One gpurun call: smoke, benches, rocprofv3 kernel-trace summary. Outputs under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python __graft_entry__.py smoke 2>&1 | tail -5
nproc
python bench.py --workload c2 > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err; tail -c 3000 gpurun_out/bench_c2.json; tail -3 gpurun_out/bench_c2.err
python bench.py --workload c3 --no-cpu-baseline > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err; tail -c 3000 gpurun_out/bench_c3.json; tail -3 gpurun_out/bench_c3.err
REPO=$PWD
cd /tmp && rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_c2 -o c2 -- python $REPO/bench.py --workload c2 --no-cpu-baseline --steps 10 > $REPO/gpurun_out/prof_c2.log 2>&1
cd $REPO
find gpurun_out/prof_c2 -name "*stats*" | head; 
f=$(find gpurun_out/prof_c2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -20 "$f"

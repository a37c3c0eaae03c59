#!/bin/bash
# rocprofv3 kernel trace of the LDP per-frame path (bench.py --workload c5)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp; REPO=$PWD
python bench.py --workload c5 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_c5.json 2>gpurun_out/bench_c5.err || tail -3 gpurun_out/bench_c5.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_c5 -o c5 -- python $REPO/bench.py --workload c5 --steps 200 --warmup 20 --no-cpu-baseline > $REPO/gpurun_out/prof_c5.log 2>&1
cd $REPO
head -12 gpurun_out/prof_c5/c5_kernel_stats.csv | cut -c1-150
python scripts/latency_ldp.py | tail -6
python - <<'PY'
import json; d=json.load(open("gpurun_out/bench_c5.json")); print(d["value"], d["ms_per_step"], d["stages_ms_per_step"])
PY
python - <<'PY'
# gaps between the three kernels of a frame (kernel trace of the same run): end -> next start
import csv, glob
rows = []
for f in glob.glob("gpurun_out/prof_c5/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:28]))
rows.sort()
rows = rows[len(rows) // 2: len(rows) // 2 + 9]
for i, (s, e, n) in enumerate(rows):
    gap = s - rows[i - 1][1] if i else 0
    print("%-30s start +%7.1f us  dur %6.1f us  gap before %5.1f us" % (n, (s - rows[0][0]) / 1e3, (e - s) / 1e3, gap / 1e3))
PY

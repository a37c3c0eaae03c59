#!/bin/bash
# round 5: where a plan-3 step goes, per kernel: rocprofv3 kernel trace + stats of a bench run whose only fast region is plan 3, then
# (PMC=1) the HBM-side traffic of every dispatch of the step (separate FETCH_SIZE / WRITE_SIZE passes, kernel trace only).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
TAG=${TAG:-p3}
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/${TAG}_trace -o p -- python $REPO/bench.py --no-cpu-baseline --no-host-scopes --fast-plans 3 --steps 20 --warmup 3 > $REPO/gpurun_out/${TAG}_trace.json 2> $REPO/gpurun_out/${TAG}_trace.err || tail -3 $REPO/gpurun_out/${TAG}_trace.err
if [ -n "${PMC:-}" ]; then
  for pmc in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $REPO/gpurun_out/${TAG}_$pmc -o p -- python $REPO/bench.py --no-cpu-baseline --no-host-scopes --fast-plans 3 --steps 3 --warmup 1 --ramp-ms 30 > /dev/null 2>&1
  done
fi
cd $REPO
python - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/${TAG}_trace/**/*kernel_stats.csv", recursive=True)
if f:
    print("# rocprofv3 --kernel-trace --stats (bench.py --fast-plans 3 --steps 20): every ethcnn kernel")
    for r in csv.DictReader(open(f[0])):
        if "ethcnn" in r["Name"]:
            print("%-70s calls %5s  avg %9.1f us  total %8.2f ms  %5s %%" % (r["Name"].split("(")[0][-70:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
for pmc in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for fn in glob.glob("gpurun_out/${TAG}_%s/**/*counter_collection.csv" % pmc, recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"].split("(")[0][-60:]
            if "ethcnn" in r["Kernel_Name"]:
                agg[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k in sorted(agg):
        v = agg[k] / cnt[k]
        gb = v * (2048 if pmc == "FETCH_SIZE" else 1024) / 1e9  # FETCH_SIZE counts 32-B units as KB on gfx950 (x2), WRITE_SIZE KB
        print("%-10s %-60s %12.0f KB -> %.3f GB per dispatch (%d dispatches)" % (pmc, k, v, gb, cnt[k]))
PY

#!/bin/bash
# HBM-side traffic of a whole C3 step per plan: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (kernel
# trace only) of a short bench run that visits the exact plan and plans 2 / 3; per-kernel per-dispatch averages, summed over the
# dispatches of ONE step of each plan (C3: one pass per step, every kernel once) -> gpurun_out/step_traffic.json (committed as
# profiles/step_traffic.json; bench.py puts it into the line as `hbm`).  FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
for pmc in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $REPO/gpurun_out/steptraffic_$pmc -o p -- python $REPO/bench.py --no-cpu-baseline --no-host-scopes --no-other-configs --fast-plans 2,3 --steps 3 --warmup 1 --ramp-ms 30 > $REPO/gpurun_out/steptraffic_$pmc.log 2>&1 || tail -3 $REPO/gpurun_out/steptraffic_$pmc.log
done
cd $REPO
python - <<'PY'
import collections, csv, glob, json, sys
sys.path.insert(0, ".")
import bench
avg = {}
for pmc in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for fn in glob.glob("gpurun_out/steptraffic_%s/**/*counter_collection.csv" % pmc, recursive=True):
        for r in csv.DictReader(open(fn)):
            if "ethcnn" in r["Kernel_Name"] and r["Counter_Name"] == pmc:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ethcnn::", "")
                agg[k] += float(r["Counter_Value"]); cnt[k] += 1
    for k in agg:
        avg.setdefault(k, {})[pmc] = agg[k] / cnt[k]
def nbytes(k):
    d = avg[k]
    return int(d.get("FETCH_SIZE", 0.0) * 1024 * 2 + d.get("WRITE_SIZE", 0.0) * 1024)
def find(prefix):
    c = [k for k in avg if k.startswith(prefix)]
    if not c: raise SystemExit("no dispatch of %s in the PMC passes: %s" % (prefix, sorted(avg)))
    return c
plans = {"c3": ["k0_tile_slab", "k1_trunk<false, 0>", "k_fc1_bulk", "k_heads<", "k5_gate"],
         "c3_plan2": ["k0_tile_slab", "k1_trunk<false, 2>", "k_fc1_fast", "k_heads<", "k5_gate"],
         "c3_plan3": ["k1_trunk_f16_foldall", "k_fc1_fast", "k_heads_f16", "k5_gate"]}
out = {"kernel_source_stamp": bench.kernel_source_stamp(),
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (kernel trace only) of `python bench.py --no-cpu-baseline --no-host-scopes "
                 "--no-other-configs --fast-plans 2,3 --steps 3 --warmup 1` (scripts/gpu_step_traffic.sh): per-dispatch averages of every kernel, "
                 "summed over the dispatches of one C3 step of the plan; FETCH_SIZE x2 (gfx950 correction)"}
n = 102000
alg = {"c3": n * 4180, "c3_plan2": n * 4180, "c3_plan3": n * 4180}
for key, pats in plans.items():
    per = {}
    for p in pats:
        for k in find(p):
            per[k] = nbytes(k)
    out[key] = {"bytes_per_step": sum(per.values()), "per_kernel": per, "algorithmic_bytes_per_step": alg[key]}
    print(key, "%.3f GB per step" % (out[key]["bytes_per_step"] / 1e9), {k: "%.3f GB" % (v / 1e9) for k, v in per.items()})
json.dump(out, open("gpurun_out/step_traffic.json", "w"), indent=1)
PY

#!/bin/bash
# A/B of FC1 tile shapes (ETHCNN_FC1_VARIANT) on workload c3 + PMC passes.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
VARIANTS=${VARIANTS:-"0 1 2 3 4 5 6"}
PMCV=${PMCV:-"0 1"}
for v in $VARIANTS; do
  ETHCNN_FC1_VARIANT=$v python bench.py --workload c3 --no-cpu-baseline --steps 10 > gpurun_out/v$v.json 2>gpurun_out/v$v.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/v$v.json"))
    print("variant $v: value %.2fM CTU/s  fc1 %.3f ms (%.1f TF, frac %.3f)  stages %s  parity %s" % (d["value"]/1e6, d["roofline"]["avg_launch_ms"], d["roofline"]["achieved"], d["roofline"]["frac"], {k: round(x,3) for k,x in d["stages_ms_per_step"].items()}, d["parity_first_frame_bit_exact"]))
except Exception as e:
    print("variant $v failed", e); print(open("gpurun_out/v$v.err").read()[-2000:])
PY
done
cd /tmp
for v in $PMCV; do
for pmc in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  tag=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  ETHCNN_FC1_VARIANT=$v rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_v${v}_$tag -o p -- python $REPO/bench.py --workload c3 --no-cpu-baseline --steps 3 --warmup 1 > $REPO/gpurun_out/pmc_v${v}_$tag.log 2>&1
done
done
cd $REPO
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:60]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
    print(f)
    for k,d in agg.items():
        print("   ", k, {c: "%.4g" % (v/cnt[(k,c)]) for c,v in d.items()})
PY

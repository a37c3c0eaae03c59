#!/bin/bash
# PMC passes + kernel-trace durations for the FC1 plan 1 kernels (bench runs both plans): clock (GRBM_GUI_ACTIVE / duration), matrix-pipe
# busy fraction, wait breakdown, HBM-side traffic.  Env knobs (ETHCNN_LIB, ETHCNN_FC1_FAST_SHAPE) are inherited.  SETS="1 2 5 6" picks passes.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
WL=${WL:-c3}
TAG=${TAG:-pmcf}
SETS=${SETS:-"1 2"}
cd /tmp
i=0
for pmc in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  case " $SETS " in *" $i "*) ;; *) continue;; esac
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $REPO/gpurun_out/${TAG}_$i -o p -- python $REPO/bench.py --workload $WL --no-cpu-baseline --no-host-scopes --steps 3 --warmup 1 --ramp-ms 30 > $REPO/gpurun_out/${TAG}_$i.log 2>&1 || tail -3 $REPO/gpurun_out/${TAG}_$i.log
done
cd $REPO
python - <<PY
import csv, glob, collections
tot=collections.defaultdict(dict)
dur=collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/${TAG}_*/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "ethcnn" not in k: continue
        dur[k.split("(")[0][-40:]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))*1e-3)
for f in sorted(glob.glob("gpurun_out/${TAG}_*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "ethcnn" not in k: continue
        k=k.split("(")[0][-40:]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
    for k,d in agg.items():
        for c,v in d.items(): tot[k][c]=v/cnt[(k,c)]
for k,d in tot.items():
    us=sorted(dur.get(k,[0])); med=us[len(us)//2]
    print(k, " median duration under the profiler %.1f us (%d dispatches)" % (med, len(us)))
    print("    ", {c: "%.4g" % v for c,v in sorted(d.items())})
    if d.get("GRBM_GUI_ACTIVE") and med:
        print("     shader clock = %.2f GHz (GUI_ACTIVE / 8 XCDs / duration)" % (d["GRBM_GUI_ACTIVE"]/8/med*1e-3))
    if d.get("GRBM_GUI_ACTIVE") and d.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        print("     mfma_util = %.3f (MFMA_BUSY / (GUI_ACTIVE/8 * 1024 SIMDs))" % (d["SQ_VALU_MFMA_BUSY_CYCLES"]/(d["GRBM_GUI_ACTIVE"]/8*1024)))
PY

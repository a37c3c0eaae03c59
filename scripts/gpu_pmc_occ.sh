set -u
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_occ -o p -- python $REPO/bench.py --no-cpu-baseline --no-host-scopes --no-other-configs --fast-plans 3 --steps 5 --warmup 1 > $REPO/gpurun_out/pmc_occ.log 2>&1
cd $REPO
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_occ/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter(); meta={}
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:44]
        if "ethcnn" not in k: continue
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
        meta[k]={x:r.get(x) for x in ("VGPR_Count","Accum_VGPR_Count","LDS_Block_Size","Workgroup_Size","Grid_Size","Scratch_Size")}
    for k,d in agg.items():
        print(k, meta[k], {c: "%.5g" % (v/cnt[(k,c)]) for c,v in d.items()})
PY

#!/bin/bash
# the CTU-load stage folded into the bulk trunk (experiments build, ETHCNN_TILE_FOLD=1) against the shipped form
# (k0_tile_slab on a side stream beside FC1): parity first, then C3 step / stage times, then HBM-side traffic of the two forms.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
export ETHCNN_LIB=$REPO/hevc-complexity-reduction_amd/lib_exp/libethcnn.so
{
echo "# parity of the folded form (bit-exact vs the oracle; ETHCNN_SMALL=0 sends every size through the multi-launch path)"
ETHCNN_TILE_FOLD=1 ETHCNN_SMALL=0 python -m pytest tests/test_gpu_parity.py -q -x -k "frames_bit_exact or stages_bit_exact or full_size_c3 or multi_pass" -p no:cacheprovider 2>&1 | tail -2
for rep in 1 2; do for f in 0 1; do
  ETHCNN_TILE_FOLD=$f python bench.py --no-cpu-baseline --no-host-scopes --no-fast-plan --steps 30 > gpurun_out/fold_${f}_$rep.json 2> gpurun_out/fold.err
done; done
python - <<'PY'
import json, glob
for f in (0, 1):
    for rep in (1, 2):
        d = json.load(open("gpurun_out/fold_%d_%d.json" % (f, rep)))
        st = d["stages_ms_per_step"]
        print("ETHCNN_TILE_FOLD=%d run %d: %.2f M CTU/s  ms_per_step %.3f  FC1 inside the timed region %.3f ms (frac %.3f) | stages alone: tile %.3f trunk %.3f fc1 %.3f heads %.3f  parity %s"
              % (f, rep, d["value"] / 1e6, d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], st["tile"], st["trunk"], st["fc1"], st["heads"], d["parity_first_frame_bit_exact"]))
PY
cd /tmp
for f in 0 1; do for pmc in FETCH_SIZE WRITE_SIZE; do
  ETHCNN_TILE_FOLD=$f rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $REPO/gpurun_out/foldpmc_${f}_$pmc -o p -- python $REPO/bench.py --no-cpu-baseline --no-host-scopes --no-fast-plan --steps 3 --warmup 1 --ramp-ms 30 > /dev/null 2>&1
done; done
cd $REPO
python - <<'PY'
import csv, glob, collections
for f in (0, 1):
    tot = collections.defaultdict(dict)
    for pmc in ("FETCH_SIZE", "WRITE_SIZE"):
        agg = collections.defaultdict(float); cnt = collections.Counter()
        for fn in glob.glob("gpurun_out/foldpmc_%d_%s/**/*counter_collection.csv" % (f, pmc), recursive=True):
            for r in csv.DictReader(open(fn)):
                k = r["Kernel_Name"].split("(")[0]
                if "k0_tile" in k or "k1_trunk" in k:
                    agg[k] += float(r["Counter_Value"]); cnt[k] += 1
        for k in agg: tot[k][pmc] = agg[k] / cnt[k]
    for k, d in tot.items():
        print("ETHCNN_TILE_FOLD=%d  %-40s FETCH_SIZE %.0f KB (x2 on gfx950 = %.2f GB)  WRITE_SIZE %.0f KB (%.2f GB)  per launch of 102,000 CTUs"
              % (f, k[-40:], d.get("FETCH_SIZE", 0), d.get("FETCH_SIZE", 0) * 2048 / 1e9, d.get("WRITE_SIZE", 0), d.get("WRITE_SIZE", 0) * 1024 / 1e9))
PY
} > gpurun_out/tile_fold.txt 2>&1
cat gpurun_out/tile_fold.txt

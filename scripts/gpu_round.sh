#!/bin/bash
# Round evidence in one gpurun call: gpu tests, smoke, benches, rocprofv3 kernel-trace stats and the
# HBM-traffic / MFMA-busy PMC passes of the DEFAULT bench workload (c3 = 3840x2160 QP32 x 50).
# Everything lands in gpurun_out/; scripts/collect_round.sh copies the judged summaries to profiles/.  The FETCH / WRITE counter passes run before the
# benches and refresh profiles/fc1_traffic.json + profiles/step_traffic.json on the box, so that the evidence lines carry current traffic numbers.
#   SKIP_TESTS=1  skip pytest (when the call is about numbers only)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
WL=${WL:-c3}
if [ -z "${SKIP_TESTS:-}" ]; then
  python -m pytest tests -m gpu -x -q --timeout 400 2>&1 | tail -4
  python __graft_entry__.py smoke 2>&1 | tail -3
fi
bash scripts/gpu_dist_smoke.sh 2>&1 | tail -6
# the counter passes FIRST: the bench lines below carry roofline.traffic and hbm only when the committed numbers were taken at these kernel sources
# HBM traffic of FC1 (and, for the default workload, the MFMA-busy pass): c3, then the FETCH / WRITE passes of c2 as well, so that
# profiles/fc1_traffic.json carries a current stamp for both
pmc_passes() {
cd /tmp
for pmc in "${PMCS[@]}"; do
  tag=$(echo $pmc | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_${WL}_$tag -o p -- python $REPO/bench.py --workload $WL --no-cpu-baseline --no-host-scopes --steps 5 --warmup 1 > $REPO/gpurun_out/pmc_${WL}_$tag.log 2>&1
done
cd $REPO
WL=$WL python - <<'PY'
# FC1 HBM traffic per step (= per FC1 stage "launch" of bench.py: the 128x112 main dispatch + the
# remainder dispatch) from the FETCH_SIZE / WRITE_SIZE passes (KB as reported; FETCH x2 on gfx950)
import csv, glob, json, os
wl = os.environ["WL"]
n = {"c3": 102000, "c2": 25500}[wl]  # CTUs per step (bench.py WORKLOADS)
def per_step(tag, counter):
    tot, steps = 0.0, 0
    for f in glob.glob("gpurun_out/pmc_%s_%s/**/*counter_collection.csv" % (wl, tag), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_fc1" in r["Kernel_Name"] and "k_fc1_fast" not in r["Kernel_Name"] and r["Counter_Name"] == counter:
                tot += float(r["Counter_Value"])
                steps += 1 if ("k_fc1_bulk" in r["Kernel_Name"] or "<2, 7, 4, 1" in r["Kernel_Name"]) else 0
    return (tot / steps, steps) if steps else (None, 0)
(fe, n1), (wr, n2) = per_step("FETCH_SIZE","FETCH_SIZE"), per_step("WRITE_SIZE","WRITE_SIZE")
if fe is not None and wr is not None:
    import sys
    sys.path.insert(0, ".")
    import bench
    out = {wl: {"kernel_source_blob": bench.fc1_source_stamp(),
                "bytes_per_launch": int(fe*1024*2 + wr*1024), "fetch_size_kb_reported": fe, "write_size_kb_reported": wr,
                "steps_averaged": n1, "algorithmic_bytes_per_launch": n*2688*4 + 2688*448*4 + n*448*4,
                "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `python bench.py --workload %s --no-cpu-baseline --no-host-scopes --steps 5 --warmup 1` (scripts/gpu_round.sh), summed over the FC1 dispatches of a step; FETCH_SIZE x2 (gfx950 correction)" % wl}}
    # the fast plans' FC1 kernel (bench.py runs plans 2 and 3, both with k_fc1_fast<2, ..>): one dispatch per step each
    for plan in (2,):
        def fast(tag, counter):
            v = [float(r["Counter_Value"]) for f in glob.glob("gpurun_out/pmc_%s_%s/**/*counter_collection.csv" % (wl, tag), recursive=True)
                 for r in csv.DictReader(open(f)) if "k_fc1_fast<%d" % plan in r["Kernel_Name"] and r["Counter_Name"] == counter]
            return (sum(v) / len(v), len(v)) if v else (None, 0)
        (ffe, k1), (fwr, k2) = fast("FETCH_SIZE", "FETCH_SIZE"), fast("WRITE_SIZE", "WRITE_SIZE")
        if ffe is not None and fwr is not None:
            npieces = 2
            out["%s_plan%d" % (wl, plan)] = {"kernel_source_blob": bench.fc1_fast_source_stamp(),
                "bytes_per_launch": int(ffe*1024*2 + fwr*1024), "fetch_size_kb_reported": ffe, "write_size_kb_reported": fwr, "steps_averaged": k1,
                "algorithmic_bytes_per_launch": n*2688*2*npieces + 2688*448*2*npieces + n*448*4,
                "source": "the same passes, dispatches of k_fc1_fast<%d, 7, ...> (FETCH_SIZE x2)" % plan}
    json.dump(out, open("gpurun_out/fc1_traffic_%s.json" % wl,"w"), indent=1); print("fc1 traffic:", {k: v["bytes_per_launch"] for k, v in out.items()})
    # ... and into the copy bench.py reads on THIS box (profiles/fc1_traffic.json; scripts/collect_round.sh takes the same numbers home)
    try:
        cur = json.load(open("profiles/fc1_traffic.json"))
    except Exception:
        cur = {}
    cur.update(out)
    json.dump(cur, open("profiles/fc1_traffic.json", "w"), indent=1)
PY
}
PMCS=("FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA"); pmc_passes
if [ "$WL" = c3 ]; then WL=c2; PMCS=("FETCH_SIZE" "WRITE_SIZE"); pmc_passes; WL=c3; fi
# whole-step HBM traffic per plan (bench.py's `hbm` object reads profiles/step_traffic.json: refreshed here, on the box, in front of the benches)
bash scripts/gpu_step_traffic.sh 2>&1 | tail -4
[ -s gpurun_out/step_traffic.json ] && cp gpurun_out/step_traffic.json profiles/step_traffic.json
python bench.py > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err || tail -5 gpurun_out/bench_c3.err
python bench.py --workload c2 --no-cpu-baseline > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err || tail -5 gpurun_out/bench_c2.err
python bench.py --workload c4 --no-cpu-baseline --steps 5 > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err || tail -5 gpurun_out/bench_c4.err
python bench.py --workload c5 --steps 200 --warmup 20 --cpu-seconds 8 > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err || tail -5 gpurun_out/bench_c5.err
python scripts/summarize.py "gpurun_out/bench_c*.json"
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_c3.json"))
for e in d.get("cpu_baselines", []): print("cpu:", {k: e.get(k) for k in ("name", "value", "cores", "scope")})
print("host_scopes:", d.get("host_scopes"))
PY
python scripts/latency.py > gpurun_out/latency.txt 2>&1; cat gpurun_out/latency.txt
# (development knobs are read by the experiments build only: ETHCNN_LIB selects it)
EXP=$REPO/hevc-complexity-reduction_amd/lib_exp/libethcnn.so
{ echo "# the same with the single-launch small pass OFF (experiments build, ETHCNN_SMALL=0: tile / trunk / FC1 / heads / gate launches)"; ETHCNN_LIB=$EXP ETHCNN_SMALL=0 python scripts/latency.py; } > gpurun_out/latency_five_launches.txt 2>&1; cat gpurun_out/latency_five_launches.txt
python scripts/power_probe.py 2500 > gpurun_out/power_probe.txt 2>&1; tail -70 gpurun_out/power_probe.txt
python scripts/ldp_handshake.py 1000 > gpurun_out/ldp_handshake.txt 2>&1; cut -c1-200 gpurun_out/ldp_handshake.txt
python scripts/latency_ldp.py --cpu > gpurun_out/latency_ldp.txt 2>&1; cat gpurun_out/latency_ldp.txt
python scripts/latency_ldp_stream.py > gpurun_out/latency_ldp_stream.txt 2>&1; cat gpurun_out/latency_ldp_stream.txt
python scripts/latency_hook.py 2>&1 | grep -v "^ethcnn (in-process)" > gpurun_out/latency_hook.txt; cat gpurun_out/latency_hook.txt
{ python scripts/latency_host.py; echo "# --- the same with ETHCNN_PULL=0 (experiments build): copy engine first, banded above 1024 CTUs (the round's first form)"; ETHCNN_LIB=$EXP ETHCNN_PULL=0 python scripts/latency_host.py; } > gpurun_out/latency_host.txt 2>&1; cat gpurun_out/latency_host.txt
bash scripts/gpu_pull_probe.sh > /dev/null 2>&1; grep -E "^===|launch" gpurun_out/pull_timeline.txt | cut -c1-200
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$WL -o $WL -- python $REPO/bench.py --workload $WL --no-cpu-baseline --no-host-scopes > $REPO/gpurun_out/prof_$WL.log 2>&1
cd $REPO
head -12 gpurun_out/prof_$WL/${WL}_kernel_stats.csv
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_c*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:48]
        if "ethcnn" not in k: continue
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
    for k,d in agg.items():
        print(k, {c: "%.5g" % (v/cnt[(k,c)]) for c,v in d.items()})
PY

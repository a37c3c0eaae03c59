#!/bin/bash
# Round evidence in one gpurun call: gpu tests, smoke, benches, rocprofv3 kernel-trace stats and the
# HBM-traffic / MFMA-busy PMC passes of the DEFAULT bench workload (c3 = 3840x2160 QP32 x 50).
# Everything lands in gpurun_out/; scripts/collect_round.sh copies the judged summaries to profiles/.  The FETCH / WRITE counter passes run before the
# benches and refresh profiles/step_traffic.json on the box, so that the evidence lines carry current traffic numbers.
#   SKIP_TESTS=1  skip pytest (when the call is about numbers only)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$PWD
WL=${WL:-c3}
if [ -z "${SKIP_TESTS:-}" ]; then
  python -m pytest tests -m gpu -x -q --timeout 900 2>&1 | tail -6 | tee gpurun_out/gpu_tests.txt
  python __graft_entry__.py smoke 2>&1 | tail -3 | tee -a gpurun_out/gpu_tests.txt
fi
bash scripts/gpu_dist_smoke.sh 2>&1 | tail -6
# the counter passes FIRST: the bench lines below carry roofline.traffic and hbm only when the committed numbers were taken at these kernel sources
# (a) HBM traffic per kernel AND grid size, c3 and c2, each workload alone in its run -> profiles/step_traffic.json on the box
bash scripts/gpu_step_traffic.sh 2>&1 | tail -8
[ -s gpurun_out/step_traffic.json ] && cp gpurun_out/step_traffic.json profiles/step_traffic.json
# (b) SQ counters of the exact plan's kernels, the default workload alone (no other configs, no fast plans in the trace)
cd /tmp
for pmc in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA"; do
  tag=$(echo $pmc | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_${WL}_$tag -o p -- python $REPO/bench.py --workload $WL --no-cpu-baseline --no-host-scopes --no-other-configs --no-fast-plan --steps 5 --warmup 1 > $REPO/gpurun_out/pmc_${WL}_$tag.log 2>&1
done
cd $REPO
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
# kernel stats of the default workload ALONE (no other configs, no fast plans: every kernel of the trace is the exact C3 step's), then the
# plan-3 region the same way; rocprofv3's own --stats summary lumps grid sizes, scripts/kernel_stats_by_grid.py splits them
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$WL -o $WL -- python $REPO/bench.py --workload $WL --no-cpu-baseline --no-host-scopes --no-other-configs --no-fast-plan > $REPO/gpurun_out/prof_$WL.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/prof_${WL}_plan3 -o p3 -- python $REPO/bench.py --workload $WL --no-cpu-baseline --no-host-scopes --no-other-configs --fast-plans 3 > $REPO/gpurun_out/prof_${WL}_plan3.log 2>&1
cd $REPO
python scripts/kernel_stats_by_grid.py $(find gpurun_out/prof_$WL -name "*kernel_trace.csv" | head -1) gpurun_out/kernel_stats_by_grid_$WL.csv "rocprofv3 --kernel-trace --stats -- python bench.py --workload $WL --no-cpu-baseline --no-host-scopes --no-other-configs --no-fast-plan (scripts/gpu_round.sh): the exact plan of $WL alone"
python scripts/kernel_stats_by_grid.py $(find gpurun_out/prof_${WL}_plan3 -name "*kernel_trace.csv" | head -1) gpurun_out/kernel_stats_by_grid_${WL}_plan3.csv "rocprofv3 --kernel-trace -- python bench.py --workload $WL --no-cpu-baseline --no-host-scopes --no-other-configs --fast-plans 3 (scripts/gpu_round.sh): exact region, then the plan-3 region"
head -12 gpurun_out/kernel_stats_by_grid_$WL.csv
# where a step's time goes between its kernels, c3 against c2: the c2 trace the same way, then scripts/trace_gaps.py
if [ "$WL" = c3 ]; then
  cd /tmp
  rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/prof_c2 -o c2 -- python $REPO/bench.py --workload c2 --no-cpu-baseline --no-host-scopes --no-other-configs --no-fast-plan > $REPO/gpurun_out/prof_c2.log 2>&1
  cd $REPO
  python scripts/kernel_stats_by_grid.py $(find gpurun_out/prof_c2 -name "*kernel_trace.csv" | head -1) gpurun_out/kernel_stats_by_grid_c2.csv "rocprofv3 --kernel-trace -- python bench.py --workload c2 --no-cpu-baseline --no-host-scopes --no-other-configs --no-fast-plan (scripts/gpu_round.sh): the exact plan of c2 alone"
  { python scripts/trace_gaps.py $(find gpurun_out/prof_c3 -name "*kernel_trace.csv" | head -1) "c3 (3840x2160 x 50 = 102,000 CTUs per step), exact plan"
    python scripts/trace_gaps.py $(find gpurun_out/prof_c2 -name "*kernel_trace.csv" | head -1) "c2 (1920x1080 x 50 = 25,500 CTUs per step), exact plan"; } > gpurun_out/step_gaps.txt 2>&1
  cat gpurun_out/step_gaps.txt
fi
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_c*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        if "ethcnn" not in r["Kernel_Name"]: continue
        k=r["Kernel_Name"].replace("void ","").replace("ethcnn::","").split("(")[0]+" @"+r["Grid_Size"]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
    for k,d in agg.items():
        print(k, {c: "%.5g" % (v/cnt[(k,c)]) for c,v in d.items()})
PY

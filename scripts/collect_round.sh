#!/bin/bash
# copies the judged summaries of the last scripts/gpu_round.sh call from gpurun_out/ (scratch) to profiles/ (tracked)
# usage: scripts/collect_round.sh r02
set -eu
R=${1:-r06}
cd "$(dirname "$0")/.."
for wl in c2 c3 c4 c5; do [ -s gpurun_out/bench_$wl.json ] && cp gpurun_out/bench_$wl.json profiles/${R}_bench_$wl.json; done
for wl in c2 c3; do
  [ -s gpurun_out/prof_$wl/${wl}_kernel_stats.csv ] && cp gpurun_out/prof_$wl/${wl}_kernel_stats.csv profiles/${R}_rocprofv3_kernel_stats_$wl.csv
done
[ -s gpurun_out/latency.txt ] && cp gpurun_out/latency.txt profiles/${R}_latency.txt
[ -s gpurun_out/latency_ldp.txt ] && cp gpurun_out/latency_ldp.txt profiles/${R}_latency_ldp.txt
[ -s gpurun_out/latency_ldp_stream.txt ] && cp gpurun_out/latency_ldp_stream.txt profiles/${R}_latency_ldp_stream.txt
if [ -s gpurun_out/latency_hook.txt ]; then { grep "^# " profiles/${R}_latency_hook.txt 2>/dev/null | grep -v "^# ethcnn_hm_predict_picture"; cat gpurun_out/latency_hook.txt; } > /tmp/_hk.txt && cp /tmp/_hk.txt profiles/${R}_latency_hook.txt; fi
if [ -s gpurun_out/latency_host.txt ]; then { grep "^# " profiles/${R}_latency_host.txt 2>/dev/null | grep -v "^# ---"; cat gpurun_out/latency_host.txt; } > /tmp/_lh.txt && cp /tmp/_lh.txt profiles/${R}_latency_host.txt; fi
[ -s gpurun_out/pull_timeline.txt ] && cp gpurun_out/pull_timeline.txt profiles/${R}_pull_timeline.txt
if [ -s gpurun_out/row_arrival_probe.txt ]; then { grep "^#" profiles/${R}_row_arrival_probe.txt 2>/dev/null; cat gpurun_out/row_arrival_probe.txt; } > /tmp/_ra.txt && cp /tmp/_ra.txt profiles/${R}_row_arrival_probe.txt; fi
[ -s gpurun_out/lstm_timeline.txt ] && cp gpurun_out/lstm_timeline.txt profiles/${R}_lstm_timeline.txt
if [ -s gpurun_out/latency_mid_now.txt ]; then { grep "^#" profiles/${R}_latency_mid.txt 2>/dev/null; echo "--- at the round's HEAD (shape 0 up to 1536 CTUs)"; cat gpurun_out/latency_mid_now.txt; echo "--- the measurement the rule came from:"; grep -v "^#" profiles/${R}_latency_mid.txt 2>/dev/null | grep -v "at the round's HEAD" ; } > /tmp/_mid.txt && cp /tmp/_mid.txt profiles/${R}_latency_mid.txt; fi
[ -s gpurun_out/latency_five_launches.txt ] && cp gpurun_out/latency_five_launches.txt profiles/${R}_latency_five_launches.txt
[ -s gpurun_out/small_pass_timeline.txt ] && cp gpurun_out/small_pass_timeline.txt profiles/${R}_small_pass_timeline.txt
[ -s gpurun_out/launch_plans.txt ] && cp gpurun_out/launch_plans.txt profiles/${R}_launch_plans.txt
[ -s gpurun_out/dist_smoke.json ] && cp gpurun_out/dist_smoke.json profiles/${R}_dist_smoke_2ranks_1gpu.json
[ -s gpurun_out/power_probe.txt ] && cp gpurun_out/power_probe.txt profiles/${R}_power_probe.txt
if [ -s gpurun_out/ldp_handshake.txt ]; then { grep "^#" gpurun_out/ldp_handshake.txt; grep "^# native\|^# first form\|^#   \|^#    \|^# Boxes" profiles/${R}_ldp_handshake.txt 2>/dev/null; grep -v "^#" gpurun_out/ldp_handshake.txt; } > /tmp/_hs.txt && cp /tmp/_hs.txt profiles/${R}_ldp_handshake.txt; fi
[ -s gpurun_out/step_traffic.json ] && cp gpurun_out/step_traffic.json profiles/step_traffic.json
[ -s gpurun_out/step_gaps.txt ] && cp gpurun_out/step_gaps.txt profiles/${R}_step_gaps.txt
[ -s gpurun_out/cold_start.txt ] && cp gpurun_out/cold_start.txt profiles/${R}_cold_start.txt
for f in gpurun_out/traffic_by_kernel_grid_*.csv gpurun_out/kernel_stats_by_grid_*.csv; do [ -s "$f" ] && cp "$f" profiles/${R}_$(basename "$f"); done
python - "$R" <<'PY'
import csv, glob, collections, json, sys
R = sys.argv[1]
# per-kernel PMC averages of the default workload's passes
lines = []
for f in sorted(glob.glob("gpurun_out/pmc_c3_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "ethcnn" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].replace("void ", "").replace("ethcnn::", "").split("(")[0] + " @" + r["Grid_Size"]  # kernel AND grid size
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        lines.append("%-62s %s" % (k, {c: "%.5g" % (v / cnt[(k, c)]) for c, v in d.items()}))
if lines:
    open("profiles/%s_pmc_c3.txt" % R, "w").write(
        "# rocprofv3 --pmc passes (one counter set per pass, --kernel-trace only) of `python bench.py --workload c3 --no-cpu-baseline\n"
        "# --no-host-scopes --no-other-configs --no-fast-plan --steps 5 --warmup 1` (scripts/gpu_round.sh); per-dispatch averages per kernel @grid size.  FETCH_SIZE / WRITE_SIZE in KB as\n"
        "# reported (FETCH_SIZE x2 on gfx950 for bytes); SQ_* summed over the chip; GRBM_GUI_ACTIVE summed over the 8 XCDs.\n" + "\n".join(lines) + "\n")
PY
ls profiles | grep "^$R" | tr '\n' ' '; echo

#!/bin/bash
# A/B: fast plans, CTU-load stage of pass i+1 beside FC1 of pass i (shipped for the exact plan) or behind it (beside heads + gates)
set -u
mkdir -p gpurun_out
EXP=$PWD/hevc-complexity-reduction_amd/lib_exp/libethcnn.so
for v in 0 1 0 1; do
  ETHCNN_LIB=$EXP ETHCNN_TILE_AFTER_FC1=$v python bench.py --workload c3 --no-cpu-baseline --no-host-scopes > gpurun_out/taf_$v.json 2> gpurun_out/taf.err || tail -3 gpurun_out/taf.err
  python - $v <<'PY'
import json, sys
d = json.load(open("gpurun_out/taf_%s.json" % sys.argv[1]))
out = ["ETHCNN_TILE_AFTER_FC1=%s exact %.2f M" % (sys.argv[1], d["value"] / 1e6)]
for k in ("fast_plan", "fast_plan_fp16x2", "fast_plan_fp16x2_trunk"):
    f = d.get(k) or {}
    out.append("%s %.2f M (%.3f ms; fc1 in region %.3f)" % (k, f.get("value", 0) / 1e6, f.get("ms_per_step", 0), (f.get("roofline") or {}).get("avg_launch_ms", 0)))
print(" | ".join(out))
PY
done

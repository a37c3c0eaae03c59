#!/bin/bash
# round 5: plan 3 with the heads on the 16-bit pipe (k_heads_f16) against the exact heads (experiments build, ETHCNN_PLAN3_HEADS=0):
# parity of the fast plans first, then C3 step / stage times of both forms.  (Two persistent 16-bit forms were A/B'd with a longer version
# of this script and removed: profiles/r05_plan3_heads16.txt.)
set -u
mkdir -p gpurun_out
REPO=$PWD
{
python -m pytest tests/test_gpu_fast_plan.py -m gpu -x -q --timeout 600 2>&1 | tail -8
export ETHCNN_LIB=$REPO/hevc-complexity-reduction_amd/lib_exp/libethcnn.so
for rep in 1 2; do for f in 1 0; do
  ETHCNN_PLAN3_HEADS=$f python bench.py --no-cpu-baseline --no-host-scopes --fast-plans 3 --steps 30 > gpurun_out/heads16_${f}_$rep.json 2> gpurun_out/heads16.err || tail -3 gpurun_out/heads16.err
done; done
python - <<'PY'
import json
for f in (1, 0):
    for rep in (1, 2):
        d = json.load(open("gpurun_out/heads16_%d_%d.json" % (f, rep)))
        p = d["fast_plan_fp16x2_trunk"]
        print("ETHCNN_PLAN3_HEADS=%d run %d: plan 3 %.2f M CTU/s  %.3f ms/step  stages %s  max|d| %.3g flips %s | exact %.2f M" %
              (f, rep, p["value"] / 1e6, p["ms_per_step"], {k: round(v, 3) for k, v in p["stages_ms_per_step"].items()}, p.get("max_abs_vs_exact") or float("nan"), p.get("flips_vs_exact"), d["value"] / 1e6))
PY
} > gpurun_out/heads16.txt 2>&1
cat gpurun_out/heads16.txt

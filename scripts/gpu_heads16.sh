#!/bin/bash
# round 5: plan 3's heads on the 16-bit pipe: the persistent form (k_heads_f16_v2, W2 resident in registers; ETHCNN_PLAN3_HEADS=2, shipped) against
# the first form (one head of a 64-CTU tile per block, W2 chunks through an LDS ring: =1) and the exact heads (=0), experiments build:
# parity of the fast plans first, bit-identity of the two 16-bit forms, then C3 step / stage times.
set -u
mkdir -p gpurun_out
REPO=$PWD
{
python -m pytest tests/test_gpu_fast_plan.py -m gpu -x -q --timeout 600 2>&1 | tail -5
export ETHCNN_LIB=$REPO/hevc-complexity-reduction_amd/lib_exp/libethcnn.so
python - <<'PY'
import importlib, os, subprocess, sys
import numpy as np
code = r'''
import importlib, sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import bench
pkg = importlib.import_module("hevc-complexity-reduction_amd")
out = []
for (w, h, f) in ((3840, 2160, 3), (200, 136, 2), (4928, 3264, 1), (1920, 1080, 7)):
    luma = bench.synth_luma(w, h, f, seed=77)
    c = pkg.EthCnn(device=0); c.load_synthetic(1, 8.0); c.set_small_pass_launch(False); c.set_fc1_plan(3); c.set_thresholds(0.5, 0.5)
    out.append(c.predict_luma(luma, w, h, f, 32)); c.close()
np.save(sys.argv[1], np.concatenate([o.reshape(-1) for o in out]))
'''
res = {}
for form in ("1", "2"):
    fn = "gpurun_out/heads_form_%s.npy" % form
    subprocess.run([sys.executable, "-c", code, fn], check=True, env=dict(os.environ, ETHCNN_PLAN3_HEADS=form))
    res[form] = np.load(fn)
same = np.array_equal(res["1"].view(np.uint32), res["2"].view(np.uint32))
print("16-bit heads, first form vs persistent form on 4 geometries (%d outputs): %s" % (res["1"].size, "BIT-IDENTICAL" if same else "DIFFERENT, max |d| = %g" % np.abs(res["1"] - res["2"]).max()))
PY
for rep in 1 2; do for f in 2 1 0; do
  ETHCNN_PLAN3_HEADS=$f python bench.py --no-cpu-baseline --no-host-scopes --no-other-configs --fast-plans 3 --steps 30 > gpurun_out/heads16_${f}_$rep.json 2> gpurun_out/heads16.err || tail -3 gpurun_out/heads16.err
done; done
python - <<'PY'
import json
for f in (2, 1, 0):
    for rep in (1, 2):
        d = json.load(open("gpurun_out/heads16_%d_%d.json" % (f, rep)))
        p = d["fast_plan_fp16x2_trunk"]
        print("ETHCNN_PLAN3_HEADS=%d run %d: plan 3 %.2f M CTU/s  %.3f ms/step  stages %s  max|d| %.3g flips %s | exact %.2f M" %
              (f, rep, p["value"] / 1e6, p["ms_per_step"], {k: round(v, 3) for k, v in p["stages_ms_per_step"].items()}, p.get("max_abs_vs_exact") or float("nan"), p.get("flips_vs_exact"), d["value"] / 1e6))
PY
} > gpurun_out/heads16.txt 2>&1
cat gpurun_out/heads16.txt

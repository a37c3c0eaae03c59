#!/usr/bin/env python
"""Decision stability at scale (VERDICT r01 task 3): BASELINE.json configs[2]
(3840x2160 x 50 frames, QP 22/27/32/37) with both synthetic head gains through the HIP path, against the
literal-TF-order fp32 evaluation (oracle mode 1) and the float64 restatement.  Counts the outputs within
1e-6 / 1e-5 / 1e-4 of every threshold the reference ships and the thresholded decisions that differ.
Run on the GPU box: python scripts/decision_stability.py [frames] > gpurun_out/decision_stability.json"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402

import bench  # noqa: E402
import ethcnn_np as oracle  # noqa: E402
import stability  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    pkg = importlib.import_module("hevc-complexity-reduction_amd")
    w, h = 3840, 2160
    out = {"config": "All-Intra %dx%d, %d frames per case, QP 22/27/32/37, synthetic weights seed 1 with head gain 1 and 8" % (w, h, frames),
           "thresholds_shipped": {"AI Thr_info.txt": list(stability.AI_THRESHOLDS), "LDP Thr_info.txt": list(stability.LDP_THRESHOLDS)},
           "cases": []}
    tot = {"outputs": 0, "flips_vs_literal_fp32": 0, "flips_vs_float64": 0, "within_1e-06": 0, "within_1e-05": 0, "within_0.0001": 0}
    worst = [0.0, 0.0]
    for gain in (1.0, 8.0):
        blob = oracle.synth_blob(1, gain)
        c = pkg.EthCnn(device=0)
        c.load_blob(blob)
        c.set_thresholds(-1.0, -1.0)
        for qp in (22, 27, 32, 37):
            t0 = time.time()
            luma = bench.synth_luma(w, h, frames, seed=1000 + qp)
            got = c.predict_luma(luma, w, h, frames, qp)
            lit, f64 = stability.ungated_references(blob, luma, w, h, frames, qp)
            rep = stability.report(got, lit, f64)
            rep.update(qp=qp, head_gain=gain, p_min=float(got.min()), p_max=float(got.max()),
                       knife_edge_margin_vs_literal=stability.every_flip_is_a_knife_edge(got, lit),
                       knife_edge_margin_vs_float64=stability.every_flip_is_a_knife_edge(got, f64), seconds=time.time() - t0)
            out["cases"].append(rep)
            tot["outputs"] += rep["outputs"]
            tot["flips_vs_literal_fp32"] += rep["flips_vs_literal_fp32_total"]
            tot["flips_vs_float64"] += rep["flips_vs_float64_total"]
            for d in rep["thresholds"].values():
                for k in ("within_1e-06", "within_1e-05", "within_0.0001"):
                    tot[k] += d[k]
            worst[0] = max(worst[0], rep["max_abs_vs_literal_fp32"])
            worst[1] = max(worst[1], rep["max_abs_vs_float64"])
            print("gain %g qp %d: %s" % (gain, qp, {k: rep[k] for k in ("outputs", "max_abs_vs_literal_fp32", "max_abs_vs_float64",
                                                                      "flips_vs_literal_fp32_total", "flips_vs_float64_total")}), file=sys.stderr)
        c.close()
    tot["decisions_checked"] = tot["outputs"] * len(stability.ALL_THRESHOLDS)
    tot["max_abs_vs_literal_fp32"], tot["max_abs_vs_float64"] = worst
    out["total"] = tot
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

"""Decision-stability report (TEST / MEASUREMENT INFRASTRUCTURE -- never imported by the product).

north_star asks for probabilities within 1e-4 of the reference's and for bit-exact thresholded
decisions.  TensorFlow leaves its fp32 summation order unspecified, so "the reference's value" is
only defined up to re-association; what CAN be measured is how often a decision could depend on it:

  * how many outputs lie within 1e-6 / 1e-5 / 1e-4 of a threshold the reference ships
    (AI Thr_info.txt 0.5 x6; LDP Thr_info.txt 0.4 0.6 0.3 0.7 0.2 0.8) -- the knife-edge population;
  * how many thresholded decisions (p > thr, the form of net_CNN.py:175,187 and TEncCu.cpp:419-463)
    differ between the HIP path (canonical order) and (a) the oracle's mode 1 = literal TF-op order in
    plain fp32, (b) the numpy float64 restatement -- two other legal evaluations of the same graph;
  * the largest |dp| against both.

Used by scripts/decision_stability.py (full C3 x 4 QP x 2 head gains), tests/test_gpu_stability.py
and bench.py (a small sample beside the headline number).
"""
import numpy as np

import ethcnn_np as oracle

AI_THRESHOLDS = (0.5,)
LDP_THRESHOLDS = (0.4, 0.6, 0.3, 0.7, 0.2, 0.8)
ALL_THRESHOLDS = tuple(sorted(set(AI_THRESHOLDS + LDP_THRESHOLDS)))
BANDS = (1e-6, 1e-5, 1e-4)


def ungated_references(blob, luma, w, h, nframes, qp, chunk=2048):
    """-> (literal fp32 probs, float64 probs), both [nframes*nctu, 21], no gates."""
    lit = oracle.predict_frames(blob, luma, w, h, nframes, qp, -1.0, -1.0, mode=1)
    nctu = ((w + 63) // 64) * ((h + 63) // 64)
    f64 = np.empty((nframes * nctu, 21), dtype=np.float64)
    frames = np.ascontiguousarray(luma, dtype=np.uint8).reshape(nframes, h, w)
    for f in range(nframes):
        ctus = oracle.tile_frame(frames[f], w, h)
        for s in range(0, nctu, chunk):
            f64[f * nctu + s:f * nctu + min(nctu, s + chunk)] = oracle.forward64(blob, ctus[s:s + chunk], qp)["probs"]
    return lit, f64


def report(got, lit, f64, thresholds=ALL_THRESHOLDS):
    """got: the HIP path's UNGATED probabilities (float32 [n,21])."""
    got = np.asarray(got, dtype=np.float32)
    g64 = got.astype(np.float64)
    out = {"outputs": int(got.size), "max_abs_vs_literal_fp32": float(np.abs(g64 - lit.astype(np.float64)).max()),
           "max_abs_vs_float64": float(np.abs(g64 - f64).max()), "thresholds": {}}
    tot_lit = tot_64 = 0
    for t in thresholds:
        t32 = np.float32(t)
        d = np.abs(g64 - float(t32))
        flips_lit = int(((got > t32) != (lit > t32)).sum())
        flips_64 = int(((got > t32) != (f64 > float(t32))).sum())
        tot_lit += flips_lit
        tot_64 += flips_64
        out["thresholds"]["%g" % t] = {**{"within_%g" % b: int((d <= b).sum()) for b in BANDS},
                                       "flips_vs_literal_fp32": flips_lit, "flips_vs_float64": flips_64}
    out["flips_vs_literal_fp32_total"] = tot_lit
    out["flips_vs_float64_total"] = tot_64
    out["min_distance_to_a_threshold"] = float(min(np.abs(g64 - float(np.float32(t))).min() for t in thresholds))
    return out


def every_flip_is_a_knife_edge(got, other, thresholds=ALL_THRESHOLDS):
    """A decision may differ between two legal evaluations only where both values sit within their mutual
    distance of the threshold -- i.e. never by more than max|dp|.  Returns the worst offending margin
    (0.0 when there is no flip)."""
    got = np.asarray(got, dtype=np.float64)
    other = np.asarray(other, dtype=np.float64)
    worst = 0.0
    for t in thresholds:
        t = float(np.float32(t))
        flip = (got > t) != (other > t)
        if flip.any():
            worst = max(worst, float(np.maximum(np.abs(got - t), np.abs(other - t))[flip].max()))
    return worst

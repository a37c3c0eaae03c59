"""CPU: the decision-stability accounting (oracle/stability.py) on the oracle's own canonical mode --
the same report the GPU test and scripts/decision_stability.py make for the HIP path."""
import numpy as np


def test_report_on_canonical_oracle(oracle):
    import bench
    import stability
    w, h, frames, qp = 832, 480, 2, 32
    luma = bench.synth_luma(w, h, frames, seed=3)
    for gain in (1.0, 8.0):
        blob = oracle.synth_blob(1, gain)
        can = oracle.predict_frames(blob, luma, w, h, frames, qp, -1.0, -1.0, mode=0)
        lit, f64 = stability.ungated_references(blob, luma, w, h, frames, qp, chunk=64)
        rep = stability.report(can, lit, f64)
        assert rep["outputs"] == frames * 13 * 8 * 21
        assert rep["max_abs_vs_literal_fp32"] <= 1e-5 and rep["max_abs_vs_float64"] <= 1e-5
        # a flip can only sit on a knife edge: both values within max|dp| of the threshold
        assert stability.every_flip_is_a_knife_edge(can, lit) <= rep["max_abs_vs_literal_fp32"]
        assert stability.every_flip_is_a_knife_edge(can, f64) <= rep["max_abs_vs_float64"]
        for t, d in rep["thresholds"].items():
            assert d["within_1e-06"] <= d["within_1e-05"] <= d["within_0.0001"]
            assert d["flips_vs_literal_fp32"] <= d["within_1e-05"] and d["flips_vs_float64"] <= d["within_1e-05"]


def test_flip_detector():
    import stability
    a = np.array([[0.5000001, 0.2]], dtype=np.float32)
    b = np.array([[0.4999999, 0.2]], dtype=np.float64)
    assert 0.0 < stability.every_flip_is_a_knife_edge(a, b) < 1e-6
    r = stability.report(a, b.astype(np.float32), b, thresholds=(0.5,))
    assert r["flips_vs_float64_total"] == 1 and r["thresholds"]["0.5"]["within_1e-06"] == 1

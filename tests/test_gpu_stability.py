"""-m gpu: decision stability of the HIP path on BASELINE.json configs[2] geometry (3840x2160, all four QP bands,
both synthetic head gains; 10 frames per case here, the full 50 in scripts/decision_stability.py ->
profiles/r02_decision_stability.json): <= 1e-4 against the literal-TF-order fp32 and the float64 evaluations
of the graph, and every thresholded decision that differs from either of them is a knife edge (both values
within max|dp| of the threshold) -- there is no other kind of disagreement."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("gain", [1.0, 8.0])
@pytest.mark.parametrize("qp", [22, 27, 32, 37])
def test_decisions_only_differ_on_knife_edges(pkg, oracle, qp, gain):
    import bench
    import stability
    w, h, frames = 3840, 2160, 10
    luma = bench.synth_luma(w, h, frames, seed=1000 + qp)
    blob = oracle.synth_blob(1, gain)
    c = pkg.EthCnn(device=0)
    c.load_blob(blob)
    c.set_thresholds(-1.0, -1.0)  # ungated probabilities: what the thresholds are applied to
    got = c.predict_luma(luma, w, h, frames, qp)
    c.close()
    lit, f64 = stability.ungated_references(blob, luma, w, h, frames, qp)
    rep = stability.report(got, lit, f64)
    assert rep["outputs"] == frames * 2040 * 21
    assert rep["max_abs_vs_literal_fp32"] <= 1e-4 and rep["max_abs_vs_float64"] <= 1e-4  # north star's tolerance
    assert stability.every_flip_is_a_knife_edge(got, lit) <= rep["max_abs_vs_literal_fp32"]
    assert stability.every_flip_is_a_knife_edge(got, f64) <= rep["max_abs_vs_float64"]
    # and the HIP path itself equals the canonical oracle bit for bit, so its own decisions are reproducible
    can = oracle.predict_frames(blob, luma[:2], w, h, 2, qp, -1.0, -1.0, mode=0)
    assert np.array_equal(got[:2 * 2040].view(np.uint32), can.view(np.uint32))

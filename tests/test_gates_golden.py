"""The batch gates (net_CNN.py:175,187 over the fed sub-batches of video_to_cu_depth.py:61-73) against a fixture made by
an INDEPENDENT numpy evaluation (tests/gates_ref.py; generator tests/golden/gen_gates_golden.py): thresholds exactly at a
sub-batch maximum, one ulp below, a closed L1 gate with `0 > thr2` true / false, ragged sub-batch tails, two frames.
CPU: the fixture is self-consistent and the oracle reproduces it bit for bit; -m gpu: so does the HIP path."""
import os
import sys
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
import gates_ref  # noqa: E402


@pytest.fixture(scope="module")
def fx():
    import gen_gates_golden as gen
    g = np.load(os.path.join(HERE, "golden", "gates_golden.npz"))
    seed, wseed, gain, qp, nctu, nframes = g["params"]
    luma = gen.luma_strip()
    import ctu_gen
    assert ctu_gen.crc(luma) == int(g["luma_crc"])  # the regenerated pixels are the ones the fixture was made from
    return dict(g=g, luma=luma, wseed=int(wseed), gain=float(gain), qp=int(qp), nctu=int(nctu), nframes=int(nframes))


def _want(fx, i):
    g = fx["g"]
    w = gates_ref.gate_frames(g["raw"], fx["nctu"], g["thr"][i, 0], g["thr"][i, 1])
    assert zlib.crc32(w.tobytes()) == int(g["want_crc"][i])  # what the generator committed
    return w


def test_fixture_covers_the_corner_cases(fx):
    g = fx["g"]
    st = g["states"]
    assert st.shape == (len(g["thr"]), 4)
    assert {0, 1, 2, 3} <= set(st.ravel().tolist())        # open / y32 only zeroed (0 > thr2) / y16 only / both
    assert any(len(set(r.tolist())) > 1 for r in st)      # sub-batches of one run in different states
    raw = g["raw"]
    assert float(g["thr"][1, 0]) == float(raw[:1024, 0].max())  # exactly at the maximum of a sub-batch
    for i in range(len(g["thr"])):
        w = _want(fx, i)
        assert np.array_equal(w[:, 0], raw[:, 0])          # y64 is never gated
        for k, (a, b) in enumerate(((0, 1024), (1024, 1100), (1100, 2124), (2124, 2200))):
            assert (not w[a:b, 1:5].any()) == bool(st[i, k] & 1) and (not w[a:b, 5:].any()) == bool(st[i, k] & 2)


def test_oracle_gates_match_the_independent_evaluation(fx, oracle):
    g = fx["g"]
    blob = oracle.synth_blob(fx["wseed"], fx["gain"])
    for i, (t1, t2) in enumerate(g["thr"]):
        for mode in (0, 1):
            got = oracle.predict_frames(blob, fx["luma"], 64, 64 * fx["nctu"], fx["nframes"], fx["qp"], float(t1), float(t2), mode=mode)
            if mode == 0:
                assert np.array_equal(got.view(np.uint32), _want(fx, i).view(np.uint32)), (i, t1, t2)
            else:  # literal op order: other rounding of the probabilities, same gate logic on ITS ungated values
                raw1 = oracle.predict_frames(blob, fx["luma"], 64, 64 * fx["nctu"], fx["nframes"], fx["qp"], -1.0, -1.0, mode=1)
                assert np.array_equal(got.view(np.uint32), gates_ref.gate_frames(raw1, fx["nctu"], t1, t2).view(np.uint32)), (i, t1, t2)


@pytest.mark.gpu
def test_hip_gates_match_the_independent_evaluation(fx, pkg, oracle):
    g = fx["g"]
    c = pkg.EthCnn(device=0)
    c.load_blob(oracle.synth_blob(fx["wseed"], fx["gain"]))
    try:
        c.set_thresholds(-1.0, -1.0)
        raw = c.predict_luma(fx["luma"], 64, 64 * fx["nctu"], fx["nframes"], fx["qp"])
        assert np.array_equal(raw.view(np.uint32), g["raw"].view(np.uint32))  # ungated probabilities = the fixture's
        for i, (t1, t2) in enumerate(g["thr"]):
            c.set_thresholds(float(t1), float(t2))
            got = c.predict_luma(fx["luma"], 64, 64 * fx["nctu"], fx["nframes"], fx["qp"])
            assert np.array_equal(got.view(np.uint32), _want(fx, i).view(np.uint32)), (i, t1, t2)
            # the same through the device entry point as ONE multi-frame pass (gate chunks counted per frame)
            d_in, d_out = c.alloc(fx["luma"].nbytes), c.alloc(got.nbytes)
            d_in.upload(fx["luma"])
            c.predict_luma_device(d_in, 64, 64 * fx["nctu"], fx["nframes"], fx["qp"], d_out)
            c.synchronize()
            got2 = d_out.download(np.float32, got.size).reshape(got.shape)
            d_in.free()
            d_out.free()
            assert np.array_equal(got2.view(np.uint32), got.view(np.uint32)), i
    finally:
        c.close()

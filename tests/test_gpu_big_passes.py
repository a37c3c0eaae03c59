"""GPU: big (multi-launch) passes of the device entry against the oracle: bit-identical probabilities with open, closed and mixed
gates (thresholds exactly at a sub-batch maximum), a ragged last 64-CTU tile, several passes per call, a pass that starts in the
middle of a frame (sub-batch index offset c0 != 0 in the gate kernel's chunk map), and back-to-back asynchronous calls through the
pass pipeline.  (Round 3's merged launch plans -- FC1 + heads + gates as one launch, gates inside the heads launch -- were tested
here against the separate launches; they measured equal to 1 % slower and were removed in round 6: one launch sequence per plan.)"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
import gates_ref  # noqa: E402

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _strip(nctu, nframes, seed):
    """frames of 64 x (64 nctu) pixels: nctu CTUs per frame, content varied per sub-batch so that gates end up mixed"""
    import ctu_gen
    pool = ctu_gen.make_ctus(seed, 4096)
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, 4096, size=nctu * nframes)
    # every other sub-batch of 1024 draws from the low-contrast / flat classes only (low split probabilities)
    cls_ok = np.array([i for i in range(4096) if i % 8 in (2, 7)])
    for f in range(nframes):
        for k in range(0, nctu, 1024):
            if ((k // 1024) + f) % 2:
                a, b = f * nctu + k, f * nctu + min(k + 1024, nctu)
                idx[a:b] = cls_ok[rng.integers(0, cls_ok.size, size=b - a)]
    return pool[idx].reshape(nframes, nctu * 64, 64)


def _run(c, luma, nctu, nframes, qp):
    d_in, d_out = c.alloc(luma.nbytes), c.alloc(nframes * nctu * 21 * 4)
    d_in.upload(luma)
    c.predict_luma_device(d_in, 64, 64 * nctu, nframes, qp, d_out)
    c.synchronize()
    out = d_out.download(np.float32, nframes * nctu * 21).reshape(-1, 21)
    d_in.free()
    d_out.free()
    return out


def test_one_big_pass_matches_oracle_in_every_gate_state(pkg, oracle):
    nctu, nframes, qp = 1100, 68, 32            # 74,800 CTUs: one pass; 74,800 = 1168 x 64 + 48 (ragged last tile)
    blob = oracle.synth_blob(5, 2.0)
    luma = _strip(nctu, nframes, 77)
    c = pkg.EthCnn(device=0)
    c.load_blob(blob)
    try:
        c.set_thresholds(-1.0, -1.0)
        raw = _run(c, luma, nctu, nframes, qp)
        want_raw = oracle.predict_frames(blob, luma, 64, 64 * nctu, nframes, qp, -1.0, -1.0, mode=0)
        assert np.array_equal(_bits(raw), _bits(want_raw))
        st = c.stage_times()
        m64 = float(raw[:1024, 0].max())
        m32 = float(raw[1100:2124, 1:5].max())
        states = set()
        for t1, t2 in ((0.5, 0.5), (m64, 0.5), (0.55, m32), (2.0, -0.5), (2.0, 0.0), (float(np.median(raw[:, 0])), 0.6)):
            c.set_thresholds(t1, t2)
            want = gates_ref.gate_frames(raw, nctu, t1, t2)          # independent gate evaluation on the ungated values
            assert np.array_equal(_bits(_run(c, luma, nctu, nframes, qp)), _bits(want)), (t1, t2)
            for a in range(0, raw.shape[0], 1100):
                for s0, s1 in ((a, a + 1024), (a + 1024, a + 1100)):
                    states.add((not want[s0:s1, 1:5].any(), not want[s0:s1, 5:].any()))
        assert len(states) >= 3, states                                # open, half-closed and closed sub-batches all occurred
        # one launch per stage and pass
        c.set_profiling(2)
        c.reset_stage_times()
        _run(c, luma, nctu, nframes, qp)
        st = c.stage_times()
        assert st["launches"]["fc1"] == 1 and st["launches"]["heads"] == 1 and st["launches"]["gate"] == 1
        c.set_profiling(0)
    finally:
        c.close()


def test_passes_inside_one_huge_frame(pkg, oracle):
    """one frame of 170,000 CTUs with an 81,920-CTU workspace: passes of 81,920, 81,920 (starts at sub-batch 80
    of the frame) and 6,160 CTUs; mixed gates; vs the oracle's whole-frame evaluation"""
    nctu, qp = 170000, 27
    blob = oracle.synth_blob(9, 2.0)
    luma = _strip(nctu, 1, 78)
    c = pkg.EthCnn(device=0, max_ctus_per_pass=81920)
    c.load_blob(blob)
    try:
        c.set_thresholds(-1.0, -1.0)
        raw = _run(c, luma, nctu, 1, qp)
        t1 = float(raw[1024:2048, 0].max())   # exactly the maximum of a low-contrast sub-batch: closed there, open in busy ones
        c.set_thresholds(t1, 0.6)
        want = oracle.predict_frames(blob, luma, 64, 64 * nctu, 1, qp, t1, 0.6, mode=0)
        assert np.array_equal(_bits(want), _bits(gates_ref.gate_frames(raw, nctu, t1, 0.6)))
        closed = [not want[a:a + 1024, 1:5].any() for a in range(0, nctu, 1024)]
        assert any(closed) and not all(closed)
        assert any(closed[80:160]) and not all(closed[80:160])         # inside the second (offset) pass too
        got = _run(c, luma, nctu, 1, qp)
        assert np.array_equal(_bits(got), _bits(want))
    finally:
        c.close()


def test_back_to_back_async_calls(pkg, oracle):
    """what bench.py issues: consecutive asynchronous calls (pass pipeline on: the tile stage of call i+1 zeroes the OTHER
    set of gate predicates while heads + gate of call i still use their own)"""
    nctu, nframes, qp = 2040, 40, 32   # 81,600 CTUs = 3840x2160 geometry count, as a strip
    blob = oracle.synth_blob(3, 8.0)
    luma = _strip(nctu, nframes, 79)
    c = pkg.EthCnn(device=0)
    c.load_blob(blob)
    try:
        c.set_thresholds(0.5, 0.5)
        want = oracle.predict_frames(blob, luma, 64, 64 * nctu, nframes, qp, 0.5, 0.5, mode=0)
        d_in = c.alloc(luma.nbytes)
        outs = [c.alloc(want.nbytes) for _ in range(4)]
        d_in.upload(luma)
        for rep in range(3):
            for o in outs:
                c.predict_luma_device(d_in, 64, 64 * nctu, nframes, qp, o)
            c.synchronize()
            for o in outs:
                got = o.download(np.float32, want.size).reshape(want.shape)
                assert np.array_equal(_bits(got), _bits(want)), rep
        d_in.free()
        for o in outs:
            o.free()
    finally:
        c.close()

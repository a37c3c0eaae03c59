"""-m gpu: randomized geometry, concurrent contexts, lifecycle churn -- all bit-exact vs the oracle."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_random_geometry_fuzz(ctx, oracle):
    """40 seeded random (width, height, pitch, frame_stride, frames, qp, thresholds) cases: odd sizes,
    pitch > width, padded frame strides, 1-pixel-wide ragged CTUs, every QP band."""
    rng = np.random.default_rng(20260927)
    blob = oracle.synth_blob(9, 8.0)
    ctx.load_blob(blob)
    for case in range(40):
        w = int(rng.integers(1, 700))
        h = int(rng.integers(1, 500))
        if case % 8 == 0:
            w = 64 * int(rng.integers(1, 8)) + int(rng.integers(0, 2))      # exact multiple / one column over
        pitch = w + int(rng.choice([0, 0, 1, 7, 64]))
        frames = int(rng.integers(1, 4))
        stride = pitch * h + int(rng.choice([0, 0, 13, pitch * (h // 2)]))     # 4:2:0-like gap between planes
        qp = int(rng.integers(15, 46))
        thr1, thr2 = [float(x) for x in rng.choice([0.2, 0.4, 0.5, 0.6, 0.8], size=2)]
        luma = rng.integers(0, 256, size=stride * frames + 64, dtype=np.uint8)
        if case % 3 == 0:
            luma[: luma.size // 2] = luma[: luma.size // 2] // 8 + 90           # low-contrast half
        ctx.set_thresholds(thr1, thr2)
        got = ctx.predict_luma(luma, w, h, frames, qp, pitch=pitch, frame_stride=stride)
        want = oracle.predict_frames(blob, luma, w, h, frames, qp, thr1, thr2, mode=0, pitch=pitch, frame_stride=stride)
        assert np.array_equal(_bits(got), _bits(want)), "case %d: %dx%d pitch %d stride %d frames %d qp %d" % (
            case, w, h, pitch, stride, frames, qp)
        if case % 5 == 0:  # the LDP front end on the same plane
            gv = ctx.resi_vectors(luma[: pitch * h], w, h, pitch=pitch)
            wv = oracle.resi_vectors(blob, luma[: pitch * h], w, h, pitch=pitch)
            assert np.array_equal(_bits(gv), _bits(wv)), "resi case %d" % case
    ctx.set_thresholds(0.5, 0.5)


def test_concurrent_contexts(pkg, oracle):
    """One context per thread (the ABI's threading contract): 4 threads, own contexts, different
    weights and geometries, running at the same time on the same GPU."""
    cases = [(11, 416, 240, 3, 22), (12, 200, 136, 5, 27), (13, 832, 480, 2, 32), (14, 72, 72, 9, 37)]
    results, errors = {}, []

    def work(seed, w, h, frames, qp):
        try:
            rng = np.random.default_rng(seed)
            blob = oracle.synth_blob(seed, 8.0)
            luma = rng.integers(0, 256, size=(frames, h, w), dtype=np.uint8)
            c = pkg.EthCnn(device=0)
            try:
                c.load_blob(blob)
                outs = [c.predict_luma(luma, w, h, frames, qp) for _ in range(6)]
            finally:
                c.close()
            results[seed] = (blob, luma, outs)
        except Exception as exc:  # surfaced in the main thread
            errors.append(exc)

    threads = [threading.Thread(target=work, args=c) for c in cases]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for seed, w, h, frames, qp in cases:
        blob, luma, outs = results[seed]
        want = oracle.predict_frames(blob, luma, w, h, frames, qp, 0.5, 0.5, mode=0)
        for o in outs:
            assert np.array_equal(_bits(o), _bits(want))


def test_lifecycle_churn(pkg, oracle):
    """create / load / predict / reload other weights / destroy, 25 times: no leak-driven failure,
    no stale weights."""
    rng = np.random.default_rng(3)
    luma = rng.integers(0, 256, size=(136, 200), dtype=np.uint8)
    blobs = [oracle.synth_blob(s, 4.0) for s in (1, 2)]
    wants = [oracle.predict_frames(b, luma, 200, 136, 1, 30, 0.5, 0.5, mode=0) for b in blobs]
    for i in range(25):
        c = pkg.EthCnn(device=0, max_ctus_per_pass=64 if i % 2 else 0)
        c.load_blob(blobs[i % 2])
        assert np.array_equal(_bits(c.predict_luma(luma, 200, 136, 1, 30)), _bits(wants[i % 2]))
        c.load_blob(blobs[1 - i % 2])
        assert np.array_equal(_bits(c.predict_luma(luma, 200, 136, 1, 30)), _bits(wants[1 - i % 2]))
        c.close()

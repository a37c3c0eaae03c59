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


def test_host_staging_fills_at_odd_alignments(ctx, oracle, tmp_path):
    """The staging fills write pinned memory with non-temporal 16-byte stores (head / tail bytes by memcpy): frames whose
    rows and bands start at every alignment (odd widths, odd pitches, a source buffer offset by 1..15 bytes), several
    ~512 KiB bands per frame and several frames per ring slot, through the host-memory entry point and through both forms of
    the file entry point (bounce buffer + non-temporal copy, and `pread` straight into the staging buffer)."""
    blob = oracle.synth_blob(4, 8.0)
    ctx.load_blob(blob)
    ctx.set_thresholds(0.5, 0.5)
    rng = np.random.default_rng(77)
    for k, (w, h, pad, off, frames) in enumerate([(1283, 1031, 0, 0, 3), (1283, 1031, 5, 3, 2), (2049, 777, 1, 15, 2),
                                                  (997, 2051, 63, 7, 2), (16, 16, 0, 1, 5), (4099, 515, 13, 9, 1)]):
        pitch = w + pad
        stride = pitch * h + 31
        raw = rng.integers(0, 256, size=stride * frames + 64, dtype=np.uint8)
        luma = raw[off:]                                   # a source pointer at an odd address
        got = ctx.predict_luma(luma, w, h, frames, 30, pitch=pitch, frame_stride=stride)
        want = oracle.predict_frames(blob, luma, w, h, frames, 30, 0.5, 0.5, mode=0, pitch=pitch, frame_stride=stride)
        assert np.array_equal(_bits(got), _bits(want)), "host case %d: %dx%d pitch %d offset %d" % (k, w, h, pitch, off)
    # file path: even sizes (4:2:0), rows at odd multiples of the width
    w, h, frames = 1282, 1030, 4
    planes = rng.integers(0, 256, size=(frames, h, w), dtype=np.uint8)
    yuv = tmp_path / "odd.yuv"
    with open(yuv, "wb") as f:
        for k in range(frames):
            f.write(planes[k].tobytes())
            f.write(bytes([k]) * (w * h // 2))
    want = oracle.predict_frames(blob, planes, w, h, frames, 30, 0.5, 0.5, mode=0)
    out = tmp_path / "cu_depth.dat"
    assert ctx.predict_yuv_file(str(yuv), w, h, 30, str(out)) == frames
    assert np.array_equal(_bits(np.fromfile(out, dtype=np.float32)), _bits(want).reshape(-1))


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


def test_pass_pipeline_back_to_back_async_calls(pkg, oracle):
    """The pass pipeline (tile stage of pass i+1 beside FC1 of pass i, double-buffered tile outputs / h1 / gate flags):
    eight asynchronous device calls with eight DIFFERENT inputs, geometries and QPs issued back to back without a
    synchronisation in between, then one ethcnn_synchronize: every output bit-exact vs the oracle, pipeline on and off,
    and with a workspace so small that every call is several passes."""
    rng = np.random.default_rng(2024)
    blob = oracle.synth_blob(9, 8.0)
    # passes of >= 8192 CTUs take the pipelined path (tile stage on the side stream), smaller ones stay on the main stream:
    # both kinds, interleaved
    cases = [(832, 480, 3, 32), (1920, 1080, 20, 22), (200, 136, 5, 37), (3840, 2160, 5, 27), (64 * 45, 64 * 30, 2, 27),
             (1920, 1080, 17, 32), (416, 240, 7, 32), (1280, 720, 3, 30)]
    lumas = [rng.integers(0, 256, size=(f, h, w), dtype=np.uint8) for (w, h, f, _) in cases]
    for lu in lumas:
        lu[:, : lu.shape[1] // 2] = (lu[:, : lu.shape[1] // 2] // 16 + 60).astype(np.uint8)
    wants = [oracle.predict_frames(blob, lu, w, h, f, qp, 0.5, 0.5, mode=0) for lu, (w, h, f, qp) in zip(lumas, cases)]
    for pipeline, cap in ((True, 0), (False, 0), (True, 1024), (True, 4096)):
        c = pkg.EthCnn(device=0, max_ctus_per_pass=cap)
        c.load_blob(blob)
        c.set_pass_pipeline(pipeline)
        d_in = [c.alloc(lu.nbytes) for lu in lumas]
        d_out = [c.alloc(wt.nbytes) for wt in wants]
        for b, lu in zip(d_in, lumas):
            b.upload(lu)
        for _ in range(2):  # twice: the second round starts with buffers of the first still "two passes back"
            for b_in, b_out, (w, h, f, qp) in zip(d_in, d_out, cases):
                c.predict_luma_device(b_in, w, h, f, qp, b_out)  # no synchronisation between the calls
        c.synchronize()
        for k, (b_out, wt) in enumerate(zip(d_out, wants)):
            got = b_out.download(np.float32, wt.size).reshape(wt.shape)
            assert np.array_equal(got.view(np.uint32), wt.view(np.uint32)), "pipeline=%s cap=%d case %d" % (pipeline, cap, k)
        for b in d_in + d_out:
            b.free()
        c.close()


def test_ldp_calls_between_pipelined_passes(pkg, oracle):
    """main-stream users of the workspace (LDP front-end) between pipelined All-Intra calls: the serial sections wait for
    the side stream and the next tile stage waits for them"""
    rng = np.random.default_rng(5)
    blob = oracle.synth_blob(3, 1.0)
    c = pkg.EthCnn(device=0)
    c.load_blob(blob)
    w, h, nf = 1280, 720, 36   # 36 x 240 = 8640 CTUs per call: above the 8192-CTU threshold, so the tile stage of every
                               # call really runs on the side stream (4 frames = 960 CTUs never left the main stream)
    luma = rng.integers(0, 256, size=(nf, h, w), dtype=np.uint8)
    resi = np.clip(np.rint(128 + rng.laplace(0, 6, size=(h, w))), 0, 255).astype(np.uint8)
    want_ai = oracle.predict_frames(blob, luma, w, h, nf, 32, 0.5, 0.5, mode=0)
    want_vec = oracle.resi_vectors(blob, resi, w, h, mode=0)
    d_in, d_out = c.alloc(luma.nbytes), c.alloc(want_ai.nbytes)
    d_in.upload(luma)
    for _ in range(3):
        c.predict_luma_device(d_in, w, h, nf, 32, d_out)
        vec = c.resi_vectors(resi, w, h)                    # serial section right behind an unsynchronised pipelined pass
        c.predict_luma_device(d_in, w, h, nf, 32, d_out)     # and a pipelined pass right behind it
        assert np.array_equal(vec.view(np.uint32), want_vec.view(np.uint32))
    c.synchronize()
    got = d_out.download(np.float32, want_ai.size).reshape(want_ai.shape)
    assert np.array_equal(got.view(np.uint32), want_ai.view(np.uint32))
    d_in.free()
    d_out.free()
    c.close()

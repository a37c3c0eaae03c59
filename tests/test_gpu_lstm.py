"""-m gpu: ETH-LSTM one step + LDP heads (k_lstm, through the C ABI) against oracle_lstm_step
(canonical mode): probabilities, gates and the (c, h) state bit-exact; <= 1e-4 against the
float64 restatement; on synthetic weights and on the reference's trained qp32 LSTM weights."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REAL = os.path.join(GOLDEN, "model_LDP_200000_qp32.dat")


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _inputs(rng, n, scale=1.0):
    vec = (np.abs(rng.standard_normal((n, 448))) * scale).astype(np.float32)
    vec[:, ::7] *= -0.2
    state = np.stack([rng.uniform(-5, 5, (n, 448)), rng.uniform(-1, 1, (n, 448))], 1).astype(np.float32)
    return vec, state


@pytest.fixture(scope="module")
def lstm(oracle):
    import ethcnn_lstm_np
    return ethcnn_lstm_np


@pytest.mark.parametrize("n,gain,qp,i_frame,with_state", [(1, 1.0, 32, 1, False), (16, 2.0, 22, 2, True),
                                                          (45, 4.0, 27, 7, True), (510, 2.0, 37, 4, True),
                                                          (1064, 6.0, 32, 9, True)])
def test_lstm_step_bit_exact(ctx, lstm, n, gain, qp, i_frame, with_state):
    rng = np.random.default_rng(n)
    blob = lstm.synth_lstm_blob(3, gain)
    ctx.load_lstm_blob(blob)
    ctx.set_thresholds(0.5, 0.5)
    vec, state = _inputs(rng, n)
    sin = state if with_state else None
    got_p, got_s = ctx.lstm_step(vec, sin, qp, i_frame)
    want_p, want_s = lstm.lstm_step(blob, vec, sin, qp, i_frame, 0.5, 0.5, mode=0)
    assert np.array_equal(_bits(got_s), _bits(want_s)), "state: max |d| = %g" % np.abs(got_s - want_s).max()
    assert np.array_equal(_bits(got_p), _bits(want_p)), "probs: max |d| = %g" % np.abs(got_p - want_p).max()
    ctx.set_thresholds(-1.0, -1.0)  # gates open: raw probabilities against float64
    raw_p, _ = ctx.lstm_step(vec, sin, qp, i_frame)
    rP, rS = lstm.lstm_forward64(blob, vec, sin, qp, i_frame)
    assert np.abs(raw_p - rP).max() <= 1e-4 and np.abs(got_s - rS).max() <= 1e-4
    ctx.set_thresholds(0.5, 0.5)


def test_lstm_gates_are_applied_inside_the_heads_launch(ctx, lstm):
    """The tf.cond zero-fill (net():305,317) is done by the last block of the LSTM heads launch, whose predicate words
    and ticket counter persist between launches: frame sizes with 1, 2 and 4 sub-batches in a mixed order, thresholds
    that close the L1 gate, only the L2 gate, neither, and the `0 > thr2` corner -- every call bit-exact vs the oracle,
    and every call repeated (the words a launch leaves behind must be clean)."""
    blob = lstm.synth_lstm_blob(5, 3.0)
    ctx.load_lstm_blob(blob)
    cases = [(700, 0.5, 0.5), (2040, 0.99, 0.5), (3927, 0.5, 0.999), (64, 1.5, -0.5), (2040, 0.2, 0.2), (1025, 1.5, 0.5),
             (3927, 0.7, 0.6), (16, 0.5, 1.5), (1024, 0.999, 0.999)]
    closed = 0
    for k, (n, t1, t2) in enumerate(cases):
        rng = np.random.default_rng(100 + k)
        vec, state = _inputs(rng, n)
        # make the sub-batches differ: a quiet first half raises fewer predicates
        vec[: n // 2] *= 0.25
        ctx.set_thresholds(t1, t2)
        want_p, want_s = lstm.lstm_step(blob, vec, state, 32, k + 2, t1, t2, mode=0)
        closed += int((want_p[:, 1:] == 0).all(axis=0).any() or (want_p[:, 1:5] == 0).any())
        for rep in range(2):
            got_p, got_s = ctx.lstm_step(vec, state, 32, k + 2)
            assert np.array_equal(_bits(got_p), _bits(want_p)), "case %d rep %d: %d CTUs thr %.3f / %.3f" % (k, rep, n, t1, t2)
            assert np.array_equal(_bits(got_s), _bits(want_s))
    assert closed >= 3, "the cases must actually close gates"
    ctx.set_thresholds(0.5, 0.5)


def test_lstm_synthetic_generator_matches(ctx, lstm):
    ctx.load_lstm_synthetic(17, 3.0)
    assert np.array_equal(_bits(ctx.get_lstm_blob()), _bits(lstm.synth_lstm_blob(17, 3.0)))


def test_lstm_trained_weights_recurrence(ctx, lstm):
    """the reference's own trained LSTM checkpoint, loaded through the bundle reader; 6 recurrent
    steps, each bit-exact against the oracle fed the same state"""
    ctx.load_lstm_checkpoint(REAL)
    blob = ctx.get_lstm_blob()
    ctx.set_thresholds(0.5, 0.5)
    rng = np.random.default_rng(1)
    n = 135
    g_state = o_state = None
    for i_frame in range(1, 7):
        vec, _ = _inputs(rng, n, 0.5)
        gp, g_state = ctx.lstm_step(vec, g_state, 32, i_frame)
        op, o_state = lstm.lstm_step(blob, vec, o_state, 32, i_frame, 0.5, 0.5, mode=0)
        assert np.array_equal(_bits(gp), _bits(op)) and np.array_equal(_bits(g_state), _bits(o_state)), i_frame
    assert gp[:, 0].min() > 0.0 and gp.max() < 1.0


@pytest.mark.parametrize("w,h", [(416, 240), (200, 136), (1920, 1080)])
def test_ldp_predict_frame(ctx, oracle, lstm, w, h):
    """the whole per-frame call: residual luma -> resi_cnn -> LSTM step -> heads -> gates"""
    rng = np.random.default_rng(w)
    cblob = oracle.synth_blob(21, 1.0)
    lblob = lstm.synth_lstm_blob(22, 3.0)
    ctx.load_blob(cblob)
    ctx.load_lstm_blob(lblob)
    ctx.set_thresholds(0.5, 0.5)
    g_state = o_state = None
    for i_frame in (1, 2, 3):
        luma = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
        gp, g_state = ctx.ldp_predict_frame(luma, w, h, 32, i_frame, g_state)
        vec = oracle.resi_vectors(cblob, luma, w, h)
        op, o_state = lstm.lstm_step(lblob, vec.reshape(-1, 448), o_state, 32, i_frame, 0.5, 0.5, mode=0)
        assert np.array_equal(_bits(gp), _bits(op)) and np.array_equal(_bits(g_state), _bits(o_state)), i_frame


def test_lstm_errors(pkg, oracle):
    e = pkg.ethcnn
    c = pkg.EthCnn(device=0)
    try:
        with pytest.raises(e.EthCnnError) as ei:
            c.lstm_step(np.zeros((4, 448), np.float32), None, 32, 1)
        assert ei.value.code == -5  # ETHCNN_ERR_NOWEIGHTS
        with pytest.raises(e.EthCnnError):
            c.load_lstm_blob(np.zeros(100, np.float32))
        c.load_lstm_synthetic(1, 1.0)
        with pytest.raises(e.EthCnnError) as ei:  # LSTM present, CNN absent
            c.ldp_predict_frame(np.zeros((64, 64), np.uint8), 64, 64, 32, 1)
        assert ei.value.code == -5
    finally:
        c.close()


def test_ldp_daemon_handshake(pkg, oracle, lstm, tmp_path, monkeypatch):
    """The file handshake of TEncGOP.cpp:1463-1503 played from the test as "HM": resi.yuv +
    command.dat + pred_start.sig -> wait pred_end.sig -> cu_depth.dat; 4 frames with the
    reference's trained LSTM weights (QP 32 band) and seeded CNN weights; every frame
    bit-exact against the oracle chain.  The recurrent state stays resident in HBM between frames
    (state.dat is refreshed after pred_end.sig, for protocol compatibility); before frame 4 the test
    REPLACES state.dat, which the daemon must notice and use, as the reference would."""
    import shutil
    import threading
    import time
    d = pkg.resi_to_cu_depth_LDP
    w, h, qp = 416, 240, 32
    for ext in (".index", ".data-00000-of-00001"):
        shutil.copy(REAL + ext, tmp_path / ("model_LDP_200000_qp32.dat" + ext))
    (tmp_path / "Thr_info.txt").write_text("0.4 0.6 0.3 0.7 0.2 0.8")
    monkeypatch.setenv("ETHCNN_SYNTHETIC_SEED", "21")
    result = {}
    th = threading.Thread(target=lambda: result.setdefault("n", d.serve(str(tmp_path), max_frames=4, idle_timeout=60.0,
                                                                       verbose=False)))
    th.start()
    cblob = oracle.synth_blob(21, 1.0)
    lblob = np.fromfile(REAL + ".data-00000-of-00001", dtype=np.float32)
    rng = np.random.default_rng(8)
    n = ((w + 63) // 64) * ((h + 63) // 64)
    o_state = None
    try:
        for poc in (1, 2, 3, 4):
            if poc == 4:  # somebody else's state.dat: the resident state must not be used
                o_state = (o_state * np.float32(0.5)).astype(np.float32)
                time.sleep(0.01)
                o_state.tofile(tmp_path / "state.dat")
            luma = rng.integers(96, 160, size=(h, w), dtype=np.uint8)  # residual + 128, roughly
            with open(tmp_path / "resi.yuv", "wb") as f:
                f.write(luma.tobytes())
                f.write(bytes(w * h // 2))
            (tmp_path / "command.dat").write_text("%d %d %d %d [end]" % (poc, w, h, qp))
            open(tmp_path / "pred_start.sig", "w").close()
            t0 = time.time()
            while not (tmp_path / "pred_end.sig").exists():
                assert time.time() - t0 < 60 and th.is_alive(), "daemon did not answer"
                time.sleep(0.001)
            os.remove(tmp_path / "pred_end.sig")
            assert not (tmp_path / "pred_start.sig").exists()
            got = np.fromfile(tmp_path / "cu_depth.dat", dtype=np.float32).reshape(n, 21)
            vec = oracle.resi_vectors(cblob, luma, w, h)
            want, o_state = lstm.lstm_step(lblob, vec, o_state, qp, poc, 0.6, 0.7, mode=0)
            assert np.array_equal(_bits(got), _bits(want)), poc
            t0 = time.time()  # state.dat follows the ending signal (renamed into place, so never partial)
            while True:
                try:
                    st = np.fromfile(tmp_path / "state.dat", dtype=np.float32)
                except OSError:
                    st = np.zeros(0, np.float32)
                if st.size == n * 896 and np.array_equal(_bits(st.reshape(n, 2, 448)), _bits(o_state)):
                    break
                assert time.time() - t0 < 10, "state.dat was not refreshed with frame %d's state" % poc
                time.sleep(0.001)
    finally:
        th.join(timeout=90)
    assert result.get("n") == 4


def test_ldp_step_keeps_the_state_resident(pkg, oracle, lstm):
    """ethcnn_ldp_step: state_in None = the previous call's state in HBM.  The chain equals the host-state chain
    (ethcnn_ldp_predict_frame) bit for bit; a host state overrides the resident one; a frame > 1 without a resident
    state, or after a geometry change, is an error (never silent zeros)."""
    e = pkg.ethcnn
    rng = np.random.default_rng(77)
    w, h = 832, 480
    cblob, lblob = oracle.synth_blob(21, 1.0), lstm.synth_lstm_blob(22, 3.0)
    a, b = pkg.EthCnn(device=0), pkg.EthCnn(device=0)
    try:
        for c in (a, b):
            c.load_blob(cblob)
            c.load_lstm_blob(lblob)
            c.set_thresholds(0.5, 0.5)
        with pytest.raises(e.EthCnnError):
            a.ldp_step(np.zeros((h, w), np.uint8), w, h, 32, 2)  # nothing resident yet
        host_state = None
        for i_frame in (1, 2, 3, 4, 5):
            luma = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
            pa = a.ldp_step(luma, w, h, 32, i_frame)
            pb, host_state = b.ldp_predict_frame(luma, w, h, 32, i_frame, host_state)
            assert np.array_equal(_bits(pa), _bits(pb)), i_frame
            assert np.array_equal(_bits(a.ldp_get_state(w, h)), _bits(host_state)), i_frame
        # host state given: it wins over the resident one
        other = (host_state * np.float32(0.25)).astype(np.float32)
        luma = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
        pa = a.ldp_step(luma, w, h, 32, 6, other)
        pb, _ = b.ldp_predict_frame(luma, w, h, 32, 6, other)
        assert np.array_equal(_bits(pa), _bits(pb))
        with pytest.raises(e.EthCnnError):  # geometry changed: the resident state belongs to another frame size
            a.ldp_step(np.zeros((240, 416), np.uint8), 416, 240, 32, 7)
        # buffers from ethcnn_host_alloc are used in place by the kernels (no copy launches): same bits
        pin = a.host_buffer(w * h)
        pprobs = a.host_buffer(pa.nbytes).view(np.float32).reshape(pa.shape)
        st = None
        for i_frame in (1, 2, 3):
            luma = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
            pin[:] = luma.reshape(-1)
            got = a.ldp_step(pin.reshape(h, w), w, h, 32, i_frame, probs_out=pprobs)
            pb, st = b.ldp_predict_frame(luma, w, h, 32, i_frame, st)
            assert got.ctypes.data == pprobs.ctypes.data  # written straight into the pinned buffer
            assert np.array_equal(_bits(got), _bits(pb)), i_frame
        a.free_host_buffers()
        # i_frame <= 1 restarts from zeros
        pa = a.ldp_step(luma, w, h, 32, 1)
        pb, _ = b.ldp_predict_frame(luma, w, h, 32, 1, None)
        assert np.array_equal(_bits(pa), _bits(pb))
    finally:
        a.close()
        b.close()


def a_has_state(c, w, h):
    try:
        c.ldp_get_state(w, h)
        return True
    except Exception:
        return False


def test_ldp_step_streamed_input(pkg, oracle, lstm):
    """ethcnn_ldp_step_begin / ethcnn_rows_ready / ethcnn_ldp_step_end: the frame's kernels are queued on a page-locked buffer that
    a filling thread is still writing, CTU row by CTU row in a scrambled order, some rows reported before begin was even called --
    bit-identical to ethcnn_ldp_step (probabilities and resident state) over a recurrence, for frame widths with and without
    16-byte rows and a height that ends in a short CTU row.  Misuse: luma outside page-locked memory, a second begin, end without
    begin -> ETHCNN_ERR_ARG.  A row that is never reported: the kernels give up after ~1 s, ethcnn_ldp_step_end fails, the GPU is not
    left hanging and the context keeps working."""
    import threading
    import time
    e = pkg.ethcnn
    rng = np.random.default_rng(77)
    cblob, lblob = oracle.synth_blob(23, 1.0), lstm.synth_lstm_blob(24, 3.0)
    a, b = pkg.EthCnn(device=0), pkg.EthCnn(device=0)
    try:
        for c in (a, b):
            c.load_blob(cblob)
            c.load_lstm_blob(lblob)
            c.set_thresholds(0.5, 0.5)
        for (w, h) in ((1920, 1080), (424, 240), (832, 480)):
            nctu, nrows = e.ctus_per_frame(w, h), (h + 63) // 64
            pin = a.host_buffer(w * h)
            pprobs = a.host_buffer(nctu * 21 * 4).view(np.float32)
            for i_frame in (1, 2, 3, 4):
                luma = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
                want = b.ldp_step(luma, w, h, 27, i_frame)
                order = rng.permutation(nrows)
                early = order[:2] if i_frame % 2 == 0 else order[:0]  # reported BEFORE begin (allowed once the previous step has ended)
                pin[:] = 0xAA                                          # (the previous frame's pixels must not be what is read)
                def put(cy):
                    pin[cy * 64 * w:min(h, cy * 64 + 64) * w] = luma[cy * 64:cy * 64 + 64].reshape(-1)
                    a.rows_ready(cy, cy + 1)
                for cy in early:
                    put(int(cy))
                def filler():
                    time.sleep(0.002)
                    for cy in order[len(early):]:
                        put(int(cy))
                        time.sleep(0.0002)
                t = threading.Thread(target=filler)
                t.start()
                a.ldp_step_begin(pin, w, h, 27, i_frame, pprobs)
                t.join()
                a.ldp_step_end()
                assert np.array_equal(_bits(pprobs.reshape(nctu, 21)), _bits(want)), (w, h, i_frame)
                assert np.array_equal(_bits(a.ldp_get_state(w, h)), _bits(b.ldp_get_state(w, h))), (w, h, i_frame)
            # the plain call still works on the same context and the same resident state
            luma = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
            assert np.array_equal(_bits(a.ldp_step(luma, w, h, 27, 5)), _bits(b.ldp_step(luma, w, h, 27, 5))), (w, h)
            a.free_host_buffers()
        # a pitched plane (rows 48 bytes apart from the next one's start): single-launch PULL form with a pitch
        w, h, pitch = 832, 480, 832 + 48
        nctu, nrows = e.ctus_per_frame(w, h), (h + 63) // 64
        pin = a.host_buffer(pitch * h)
        pprobs = a.host_buffer(nctu * 21 * 4).view(np.float32)
        for i_frame in (1, 2, 3):
            luma = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
            want = b.ldp_step(luma, w, h, 27, i_frame)
            pin[:] = 0x33
            a.ldp_step_begin(pin, w, h, 27, i_frame, pprobs, pitch=pitch)
            for cy in rng.permutation(nrows):
                for y in range(cy * 64, min(h, cy * 64 + 64)):
                    pin[y * pitch:y * pitch + w] = luma[y]
                a.rows_ready(int(cy), int(cy) + 1)
            a.ldp_step_end()
            assert np.array_equal(_bits(pprobs.reshape(nctu, 21)), _bits(want)), ("pitched", i_frame)
        a.free_host_buffers()
        # misuse
        w, h = 416, 240
        nctu = e.ctus_per_frame(w, h)
        pin = a.host_buffer(w * h)
        pprobs = a.host_buffer(nctu * 21 * 4).view(np.float32)
        with pytest.raises(e.EthCnnError):
            a.ldp_step_begin(np.zeros(w * h, np.uint8), w, h, 27, 1, pprobs)  # pageable luma
        with pytest.raises(e.EthCnnError):
            a.ldp_step_end()                                                   # nothing begun
        with pytest.raises(ValueError):
            a.rows_ready(3, 2)
        a.ldp_step_begin(pin, w, h, 27, 1, pprobs)
        with pytest.raises(e.EthCnnError):
            a.ldp_step_begin(pin, w, h, 27, 1, pprobs)                         # still open
        a.rows_ready(0, (h + 63) // 64)
        a.ldp_step_end()
        # a row that never comes
        luma = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
        pin[:] = luma.reshape(-1)
        a.ldp_step_begin(pin, w, h, 27, 1, pprobs)
        a.rows_ready(0, (h + 63) // 64 - 1)
        t0 = time.time()
        with pytest.raises(e.EthCnnError, match="never reported"):
            a.ldp_step_end()
        assert 0.5 < time.time() - t0 < 10.0
        with pytest.raises(e.EthCnnError):
            a.ldp_step(luma, w, h, 27, 2)                                      # frame 1 had no state before it: none is resident now
        assert np.array_equal(_bits(a.ldp_step(luma, w, h, 27, 1)), _bits(b.ldp_step(luma, w, h, 27, 1)))
        a.rows_ready(0, (h + 63) // 64)                                    # streamed again, right behind the failure
        a.ldp_step_begin(pin, w, h, 27, 2, pprobs)
        a.ldp_step_end()
        assert np.array_equal(_bits(pprobs.reshape(nctu, 21)), _bits(b.ldp_step(luma, w, h, 27, 2)))
        # ... and in the middle of a recurrence: the state that was resident BEFORE the failed step stays resident, so the frame can be run
        # again the plain way once its buffer is complete (what the native daemon does when a row of resi.yuv comes > 1 s late)
        luma3 = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
        pin[:] = luma3.reshape(-1)
        a.ldp_step_begin(pin, w, h, 27, 3, pprobs)
        a.rows_ready(1, (h + 63) // 64)
        with pytest.raises(e.EthCnnError, match="never reported"):
            a.ldp_step_end()
        want3 = b.ldp_step(luma3, w, h, 27, 3)
        assert np.array_equal(_bits(a.ldp_step(luma3, w, h, 27, 3)), _bits(want3))
        assert np.array_equal(_bits(a.ldp_get_state(w, h)), _bits(b.ldp_get_state(w, h)))
        # a streamed step that was given an explicit state_in at ANOTHER geometry gives up.  The error is the distinct
        # "rows never came" code (-7; everything else is final), what stays resident is the copy of state_in FOR ITS OWN CTU COUNT: a
        # resident step at the old geometry is refused (it used to pass the size check and read a state of the wrong size), the
        # frame itself can be run again with the same arguments
        w2, h2 = 832, 480
        n2 = e.ctus_per_frame(w2, h2)
        pin2 = a.host_buffer(w2 * h2)
        pp2 = a.host_buffer(n2 * 21 * 4).view(np.float32)
        st_in = (rng.standard_normal((n2, 2, 448)) * 0.1).astype(np.float32)
        luma4 = rng.integers(0, 256, size=(h2, w2), dtype=np.uint8)
        pin2[:] = luma4.reshape(-1)
        a.ldp_step_begin(pin2, w2, h2, 27, 2, pp2, state_in=st_in)
        a.rows_ready(0, (h2 + 63) // 64 - 1)
        with pytest.raises(e.EthCnnError, match="never reported") as ei:
            a.ldp_step_end()
        assert ei.value.code == -7
        with pytest.raises(e.EthCnnError, match="none is resident"):
            a.ldp_step(luma3, w, h, 27, 4)                                     # geometry A: its state is gone, and the library says so
        assert np.array_equal(_bits(a.ldp_get_state(w2, h2)), _bits(st_in))    # resident: the caller's state, for geometry B
        want4 = b.ldp_step(luma4, w2, h2, 27, 2, state_in=st_in)
        assert np.array_equal(_bits(a.ldp_step(luma4, w2, h2, 27, 2)), _bits(want4))   # resident step at geometry B = the same frame again
        assert np.array_equal(_bits(a.ldp_get_state(w2, h2)), _bits(b.ldp_get_state(w2, h2)))
        # zeros as the input (i_frame <= 1) while a state of another geometry is resident: the failed step wrote ONE of the two state
        # buffers -- both parities: the resident state is either still there bit for bit, or declared gone; never a state of the wrong size
        kept = 0
        for i_frame in (3, 4):
            a.ldp_step(luma4, w2, h2, 27, i_frame, state_in=None if a_has_state(a, w2, h2) else st_in)
            keep = a.ldp_get_state(w2, h2)
            a.ldp_step_begin(pin, w, h, 27, 1, pprobs)
            with pytest.raises(e.EthCnnError, match="never reported"):
                a.ldp_step_end()
            try:
                assert np.array_equal(_bits(a.ldp_get_state(w2, h2)), _bits(keep))
                kept += 1
            except e.EthCnnError:
                pass
        assert kept == 1                                                       # (one parity keeps it, the other had to drop it)
    finally:
        a.close()
        b.close()


def test_one_launch_frame_kernel_both_forms_and_forced_claim_or_execute(pkg, oracle, lstm):
    """An LDP frame's LSTM cells and heads run as ONE dataflow launch up to 640 CTUs (k_lstm_frame: heads blocks wait for the cells
    of their group and level inside the grid) and as two launches above, or with ETHCNN_LSTM_ONE_LAUNCH=0.  Both forms, frame
    sizes on both sides of the limit and of the CG = 1 / 2 switch, closed and open gates, six-frame recurrences on the resident
    state: bit-exact vs the oracle.  ETHCNN_LSTM_STEAL_TEST=k makes every k-th cell block leave WITHOUT claiming its item and
    gives the heads blocks no patience, so they must execute those cell items themselves (the forward-progress path of a
    shared GPU).  One subprocess per setting (the knobs are read once per process)."""
    import os
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import importlib, os, sys
        import numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "oracle"))
        import ethcnn_np as oracle, ethcnn_lstm_np as ol
        pkg = importlib.import_module("hevc-complexity-reduction_amd")
        c = pkg.EthCnn(0)
        bits = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
        rng = np.random.default_rng(5)
        for n, gain in ((7, 1.0), (104, 3.0), (192, 2.0), (510, 4.0), (640, 2.0), (656, 2.0), (1064, 5.0)):
            blob = ol.synth_lstm_blob(11 + n, gain)
            c.load_lstm_blob(blob)
            state = None
            want_state = None
            for i_frame in range(1, 7):
                thr = (0.5, 0.5) if i_frame %% 2 else (0.97, 0.6)
                c.set_thresholds(*thr)
                vec = (np.abs(rng.standard_normal((n, 448))) * 0.5).astype(np.float32)
                vec[:, ::5] *= -0.3
                got_p, state = c.lstm_step(vec, state, 27, i_frame)
                want_p, want_state = ol.lstm_step(blob, vec, want_state, 27, i_frame, thr[0], thr[1], mode=0)
                assert np.array_equal(bits(got_p), bits(want_p)), (n, i_frame)
                assert np.array_equal(bits(state), bits(want_state)), (n, i_frame)
        print("ok")
    """ % (root, root))
    for env in ({}, {"ETHCNN_LSTM_ONE_LAUNCH": "0"}, {"ETHCNN_LSTM_STEAL_TEST": "2"}, {"ETHCNN_LSTM_STEAL_TEST": "3"},
                {"ETHCNN_LSTM_STEAL_TEST": "7"}):
        from conftest import exp_env  # the knobs exist in the experiments build only
        r = subprocess.run([sys.executable, "-c", code], env=exp_env(**env), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "ok" in r.stdout, (env, r.stdout[-300:], r.stderr[-1500:])


def test_four_processes_run_ldp_frames_on_one_gpu(pkg, oracle, lstm, tmp_path):
    """Several encoders sharing one GPU: four processes that start together and each run LDP frames (front-end as one launch, cells +
    heads as one launch) for 2 s.  The dataflow launches' waiting blocks must not starve each other's producers (claim or
    execute): every process keeps its rate, every checked frame is bit-exact, none traps."""
    import os
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import importlib, os, sys, time
        import numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "oracle"))
        import ethcnn_np as oracle, ethcnn_lstm_np as ol
        pkg = importlib.import_module("hevc-complexity-reduction_amd")
        seed, gate = int(sys.argv[1]), sys.argv[2]
        rng = np.random.default_rng(seed)
        blob, lblob = oracle.synth_blob(6, 1.0), ol.synth_lstm_blob(4 + seed, 3.0)
        c = pkg.EthCnn(0)
        c.load_blob(blob); c.load_lstm_blob(lblob); c.set_thresholds(0.6, 0.7)
        w, h = (1920, 1080) if seed %% 2 else (832, 480)
        frames = [np.clip(np.rint(128 + rng.laplace(0, 7, size=(h, w))), 0, 255).astype(np.uint8) for _ in range(4)]
        vecs = [oracle.resi_vectors(blob, f, w, h) for f in frames]
        open(os.path.join(gate, "ready%%d" %% seed), "w").close()
        while len(os.listdir(gate)) < 4:
            time.sleep(0.001)
        t0, k, state, ostate = time.time(), 0, None, None
        while time.time() - t0 < 2.0:
            i = k %% 4
            probs, state = c.ldp_predict_frame(frames[i], w, h, 32, k + 1, state)
            if k < 12:   # the oracle recurrence is followed for the first frames (it is the slow side), then the GPU runs on
                want, ostate = ol.lstm_step(lblob, vecs[i], ostate, 32, k + 1, 0.6, 0.7, mode=0)
                assert np.array_equal(probs.view(np.uint32), want.view(np.uint32)), (seed, k)
                assert np.array_equal(state.view(np.uint32), ostate.view(np.uint32)), (seed, k)
            k += 1
        print("ok %%d %%d frames %%.2f s" %% (seed, k, time.time() - t0))
    """ % (root, root))
    gate = tmp_path / "gate"
    gate.mkdir()
    procs = [subprocess.Popen([sys.executable, "-c", code, str(s), str(gate)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for s in range(4)]
    for s, p in enumerate(procs):
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0 and ("ok %d" % s) in out, (s, out[-500:], err[-1500:])
        print(out.strip())
        assert int(out.split()[2]) >= 100 and float(out.split()[4]) < 6.0, out

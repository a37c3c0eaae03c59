"""CPU: host logic of the LDP daemon mirror (no GPU): command parsing, state file handling,
output layout -- the parts of resi_to_cu_depth_LDP.py that are not numerics."""
import numpy as np
import pytest


def test_get_command(pkg, tmp_path):
    d = pkg.resi_to_cu_depth_LDP
    f = tmp_path / "command.dat"
    f.write_text("7 416 240 32 [end]")  # TEncGOP.cpp:1474-1480 "%d %d %d %d [end]"
    assert d.get_command(str(f)) == (7, 416, 240, 32)
    f.write_text("7 416 240 32")        # still being written
    assert d.get_command(str(f)) == (-1, -1, -1, -1)
    f.write_text("")
    assert d.get_command(str(f)) == (-1, -1, -1, -1)
    assert d.get_command(str(tmp_path / "missing.dat")) == (-1, -1, -1, -1)


def test_images_and_state_files(pkg, tmp_path):
    d = pkg.resi_to_cu_depth_LDP
    w, h = 200, 136
    rng = np.random.default_rng(0)
    luma = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    y = tmp_path / "resi.yuv"
    y.write_bytes(luma.tobytes() + bytes(w * h // 2))
    got, nv = d.get_images_from_one_file(str(y), w, h, 64)
    assert nv == 4 * 3 and np.array_equal(got, luma)
    y.write_bytes(luma.tobytes()[:100])
    with pytest.raises(IOError):
        d.get_images_from_one_file(str(y), w, h, 64)
    s = tmp_path / "state.dat"
    assert d.get_state_in_from_one_file(str(s), nv, 0) is None
    assert d.get_state_in_from_one_file(str(s), nv, 1) is None  # zeros for i_frame <= 1 (:103-112)
    st = rng.standard_normal((nv, 1, 2, 448)).astype(np.float32)
    depth = rng.random((nv, 21)).astype(np.float32)
    d.save_cu_depth_and_state(depth, st, str(tmp_path / "cu_depth.dat"), str(s), str(tmp_path / "pred_end.sig"), nv)
    assert (tmp_path / "pred_end.sig").exists() and (tmp_path / "pred_end.sig").stat().st_size == 0
    assert np.array_equal(np.fromfile(tmp_path / "cu_depth.dat", dtype=np.float32).reshape(nv, 21), depth)
    back = d.get_state_in_from_one_file(str(s), nv, 2)
    assert back.shape == (nv, 1, 2, 448) and np.array_equal(back, st)
    with pytest.raises(IOError):
        d.get_state_in_from_one_file(str(s), nv + 1, 2)  # stale state.dat of another resolution


def test_constants(pkg):
    d = pkg.resi_to_cu_depth_LDP
    assert (d.VECTOR_LENGTH, d.LSTM_DEPTH, d.MINI_BATCH_SIZE, d.NUM_EXT_FEATURES) == (448, 1, 1024, 2)
    assert d.MODEL_CNN_FILE == "model_LDP_2000000_qp22~37.dat"


def test_state_sidecar_refuses_a_stale_state(pkg, tmp_path):
    """state.dat is refreshed AFTER pred_end.sig.  A daemon that died in between leaves an EARLIER frame's state on
    disk; the sidecar ("pending ..." before the ending signal, the plain tag once state.dat is written) makes that an error
    instead of a silently wrong recurrence.  A state.dat without sidecar (the reference daemon's) is accepted as it is, and
    so is a complete one from any earlier frame (HM does not predict every picture)."""
    d = pkg.resi_to_cu_depth_LDP
    nv, w, h = 4, 128, 128
    rng = np.random.default_rng(1)
    st = rng.standard_normal((nv, 1, 2, 448)).astype(np.float32)
    depth = rng.random((nv, 21)).astype(np.float32)
    s = tmp_path / "state.dat"
    d.save_cu_depth_and_state(depth, st, str(tmp_path / "cu_depth.dat"), str(s), str(tmp_path / "pred_end.sig"), nv, tag=(5, w, h))
    assert (tmp_path / "state.dat.idx").read_text().split() == ["5", str(w), str(h)]
    assert np.array_equal(d.get_state_in_from_one_file(str(s), nv, 6, (w, h)), st)   # the next frame
    assert np.array_equal(d.get_state_in_from_one_file(str(s), nv, 9, (w, h)), st)   # a later one (frames HM did not predict)
    with pytest.raises(IOError):
        d.get_state_in_from_one_file(str(s), nv, 6, (w, 64))                         # another geometry
    # the daemon dies between the ending signal and the state write of frame 6
    seen = {}

    def dying_fetch():
        seen["sidecar"] = (tmp_path / "state.dat.idx").read_text().split()
        raise KeyboardInterrupt
    with pytest.raises(KeyboardInterrupt):
        d.save_cu_depth_and_state(depth, dying_fetch, str(tmp_path / "cu_depth.dat"), str(s), str(tmp_path / "pred_end.sig"), nv, tag=(6, w, h))
    assert seen["sidecar"] == ["pending", "6", str(w), str(h)] and (tmp_path / "pred_end.sig").exists()
    with pytest.raises(IOError, match="stale"):
        d.get_state_in_from_one_file(str(s), nv, 7, (w, h))                          # frame 5's state must not feed frame 7
    (tmp_path / "state.dat.idx").unlink()
    assert np.array_equal(d.get_state_in_from_one_file(str(s), nv, 7, (w, h)), st)   # no sidecar: trusted, as in the reference
    d.save_cu_depth_and_state(depth, lambda: st * 2, str(tmp_path / "cu_depth.dat"), str(s), str(tmp_path / "pred_end.sig"), nv, tag=(6, w, h))
    assert np.array_equal(d.get_state_in_from_one_file(str(s), nv, 7, (w, h)), st * 2)

"""-m gpu: the drop-in command line (what HM's unchanged hook runs) end to end."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tfckpt_writer import write_bundle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAUNCHER = os.path.join(ROOT, "video_to_cu_depth.py")


def _yuv(path, w, h, frames, seed):
    rng = np.random.default_rng(seed)
    data = rng.integers(0, 256, size=frames * (w * h * 3 // 2), dtype=np.uint8)
    data.tofile(path)
    return data


def _run(cwd, argv, **env):
    e = dict(os.environ)
    e.update({k: str(v) for k, v in env.items()})
    return subprocess.run([sys.executable, LAUNCHER] + [str(a) for a in argv], cwd=str(cwd), env=e, capture_output=True, text=True)


def test_cli_with_checkpoint_files(pkg, oracle, tmp_path):
    """cwd holds Thr_info.txt + model_2000000_qp30~35.dat.{index,data-...}: the reference's
    own file contract (video_to_cu_depth.py:126-133, net_CNN.py:47)."""
    w, h, frames, qp = 416, 240, 3, 32
    yuv = _yuv(str(tmp_path / "seq.yuv"), w, h, frames, 1)
    (tmp_path / "Thr_info.txt").write_text("0.4 0.6 0.3 0.7 0.2 0.8\n")
    blob = oracle.synth_blob(21, 8.0)
    write_bundle(str(tmp_path / "model_2000000_qp30~35.dat"), [(n, np.array(v)) for n, v in oracle.tensor_views(blob).items()],
                 data_crc=pkg.ethcnn.crc32c_masked)
    r = _run(tmp_path, ["seq.yuv", w, h, qp])
    assert r.returncode == 0, r.stderr
    assert "Predicting Time:" in r.stdout
    got = np.fromfile(str(tmp_path / "cu_depth.dat"), dtype="<f4").reshape(-1, 21)
    want = oracle.predict_frames(blob, yuv, w, h, frames, qp, 0.6, 0.7, frame_stride=w * h * 3 // 2)
    assert got.shape == want.shape == (frames * 7 * 4, 21)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert not [f for f in os.listdir(str(tmp_path)) if ".tmp." in f]
    # a QP in another band looks for another model file, which is absent -> non-zero exit, old file untouched
    before = (tmp_path / "cu_depth.dat").read_bytes()
    r = _run(tmp_path, ["seq.yuv", w, h, 22])
    assert r.returncode != 0 and (tmp_path / "cu_depth.dat").read_bytes() == before


def test_cli_failures_exit_nonzero(tmp_path):
    _yuv(str(tmp_path / "seq.yuv"), 64, 64, 1, 2)
    (tmp_path / "Thr_info.txt").write_text("0.5 0.5 0.5 0.5 0.5 0.5\n")
    assert _run(tmp_path, ["seq.yuv", 64, 64, 32]).returncode != 0                       # no model, no opt-in
    assert _run(tmp_path, ["seq.yuv", 64, 48, 32], ETHCNN_SYNTHETIC_SEED=1).returncode != 0  # size % frame != 0
    assert _run(tmp_path, ["absent.yuv", 64, 64, 32], ETHCNN_SYNTHETIC_SEED=1).returncode != 0
    assert not (tmp_path / "cu_depth.dat").exists()
    os.remove(str(tmp_path / "Thr_info.txt"))
    assert _run(tmp_path, ["seq.yuv", 64, 64, 32], ETHCNN_SYNTHETIC_SEED=1).returncode != 0  # Thr_info.txt missing


@pytest.mark.parametrize("devices", ["0,0", "0,0,0,0,0,0,0,0"])
@pytest.mark.parametrize("form", ["threads", "processes", "native"])
def test_sharded_cli_is_byte_identical(oracle, tmp_path, devices, form):
    """Frame-range sharding (no collective).  One MI355X is visible here, so the G workers all use device 0: what is checked is that
    ranges + offsets reproduce the 1-worker file -- through the Python launcher's default (ONE process, a worker thread per device
    inside the library: ethcnn_predict_yuv_file_sharded), through its process-per-GPU form (ETHCNN_SHARD_PROCESSES=1) and through the
    native C tool (the same library entry, no interpreter at all)."""
    w, h, frames, qp = 832, 480, 11, 37   # 13 x 8 = 104 CTUs per frame, ragged bottom edge
    yuv = _yuv(str(tmp_path / "seq.yuv"), w, h, frames, 3)
    (tmp_path / "Thr_info.txt").write_text("0.5 0.5 0.5 0.5 0.5 0.5\n")
    r = _run(tmp_path, ["seq.yuv", w, h, qp], ETHCNN_SYNTHETIC_SEED=9, ETHCNN_HEAD_GAIN=8)
    assert r.returncode == 0, r.stderr
    single = (tmp_path / "cu_depth.dat").read_bytes()
    os.remove(str(tmp_path / "cu_depth.dat"))
    if form == "native":
        tool = os.path.join(ROOT, "hevc-complexity-reduction_amd", "bin", "video_to_cu_depth")
        r = subprocess.run([tool, "seq.yuv", str(w), str(h), str(qp)], cwd=str(tmp_path), capture_output=True, text=True,
                           env=dict(os.environ, ETHCNN_SYNTHETIC_SEED="9", ETHCNN_HEAD_GAIN="8", ETHCNN_DEVICES=devices, ETHCNN_TIMING="1"))
        assert r.returncode == 0 and "predict (%d workers)" % len(devices.split(",")) in r.stderr, r.stderr
    else:
        r = _run(tmp_path, ["seq.yuv", w, h, qp], ETHCNN_SYNTHETIC_SEED=9, ETHCNN_HEAD_GAIN=8, ETHCNN_DEVICES=devices,
                 ETHCNN_SHARD_PROCESSES="1" if form == "processes" else "0")
        assert r.returncode == 0, r.stderr
    assert (tmp_path / "cu_depth.dat").read_bytes() == single
    assert not [f for f in os.listdir(str(tmp_path)) if ".tmp." in f]      # no temp file left behind
    want = oracle.predict_frames(oracle.synth_blob(9, 8.0), yuv, w, h, frames, qp, 0.5, 0.5, frame_stride=w * h * 3 // 2)
    assert np.array_equal(np.frombuffer(single, dtype="<f4").view(np.uint32), want.reshape(-1).view(np.uint32))


def test_sharded_entry_reuses_its_workers_and_follows_the_context(pkg, oracle, tmp_path):
    """ethcnn_predict_yuv_file_sharded through the binding: the peers are created once and FOLLOW the calling context -- new weights,
    new thresholds, another plan, another worker count, fewer frames than workers -- byte-identical to the unsharded entry each time;
    a device list that does not start with the context's device is an argument error, and so is a device that does not exist."""
    e = pkg.ethcnn
    w, h, frames, qp = 416, 240, 9, 32
    yuv = _yuv(str(tmp_path / "seq.yuv"), w, h, frames, 7)
    c = pkg.EthCnn(device=0)
    try:
        for k, (seed, gain, thr, plan, devs) in enumerate(((3, 8.0, (0.5, 0.5), 0, [0, 0, 0]), (4, 2.0, (0.45, 0.6), 0, [0, 0, 0]),
                                                           (4, 2.0, (0.45, 0.6), 0, [0] * 5), (4, 2.0, (0.5, 0.5), 0, [0] * 16), (5, 1.0, (0.5, 0.5), 3, [0, 0]))):
            c.load_blob(oracle.synth_blob(seed, gain))
            c.set_thresholds(*thr)
            c.set_fc1_plan(plan)
            c.predict_yuv_file(str(tmp_path / "seq.yuv"), w, h, qp, str(tmp_path / "one.dat"))
            n = c.predict_yuv_file_sharded(devs, str(tmp_path / "seq.yuv"), w, h, qp, str(tmp_path / "many.dat"))
            assert n == frames and (tmp_path / "many.dat").read_bytes() == (tmp_path / "one.dat").read_bytes(), k
            if plan == 0:
                want = oracle.predict_frames(oracle.synth_blob(seed, gain), yuv, w, h, frames, qp, thr[0], thr[1], frame_stride=w * h * 3 // 2)
                assert np.array_equal(np.fromfile(str(tmp_path / "many.dat"), dtype="<f4").view(np.uint32), want.reshape(-1).view(np.uint32)), k
        c.set_fc1_plan(0)
        with pytest.raises(e.EthCnnError):
            c.predict_yuv_file_sharded([1, 0], str(tmp_path / "seq.yuv"), w, h, qp, str(tmp_path / "x.dat"))
        with pytest.raises(e.EthCnnError):
            c.predict_yuv_file_sharded([0, 99], str(tmp_path / "seq.yuv"), w, h, qp, str(tmp_path / "x.dat"))
        assert not (tmp_path / "x.dat").exists() and not [f for f in os.listdir(str(tmp_path)) if ".tmp." in f]
        a, b = c.startup_times()
        assert 0.0 <= a <= b < 60000.0
        # a plan the accuracy guard refuses for the loaded weights: the sharded entry says so BEFORE any worker starts, nothing is written
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import adversarial_blobs as ab
        c.load_blob(ab.cancelling_pairs(oracle, 21, 8.0, 1e6))
        c.set_fc1_plan(3)
        big = _yuv(str(tmp_path / "big.yuv"), 832, 480, 40, 9)     # 4160 CTUs: the multi-launch path, where the plans apply
        with pytest.raises(e.EthCnnError, match="refused") as ei:
            c.predict_yuv_file_sharded([0, 0], str(tmp_path / "big.yuv"), 832, 480, qp, str(tmp_path / "y.dat"))
        assert ei.value.code == e.ERR_PLAN_REFUSED and not (tmp_path / "y.dat").exists() and not [f for f in os.listdir(str(tmp_path)) if ".tmp." in f]
        c.set_fc1_plan(0)
        assert c.predict_yuv_file_sharded([0, 0], str(tmp_path / "big.yuv"), 832, 480, qp, str(tmp_path / "y.dat")) == 40
    finally:
        c.close()


def test_native_c_cli_matches(pkg, oracle, tmp_path):
    """tools/video_to_cu_depth.c (plain C99 over include/ethcnn.h): same command line and file
    contract, no Python in the loop; byte-identical cu_depth.dat."""
    tool = os.path.join(ROOT, "hevc-complexity-reduction_amd", "bin", "video_to_cu_depth")
    w, h, frames, qp = 200, 136, 2, 37
    yuv = _yuv(str(tmp_path / "seq.yuv"), w, h, frames, 5)
    (tmp_path / "Thr_info.txt").write_text("0.5 0.5 0.5 0.5 0.5 0.5")
    blob = oracle.synth_blob(4, 8.0)
    write_bundle(str(tmp_path / "model_2000000_qp35~40.dat"), [(n, np.array(v)) for n, v in oracle.tensor_views(blob).items()],
                 data_crc=pkg.ethcnn.crc32c_masked)
    r = subprocess.run([tool, "seq.yuv", str(w), str(h), str(qp)], cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(str(tmp_path / "cu_depth.dat"), dtype="<f4").reshape(-1, 21)
    want = oracle.predict_frames(blob, yuv, w, h, frames, qp, 0.5, 0.5, frame_stride=w * h * 3 // 2)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # failures: missing model for another band, bad size -> exit 1, message on stderr
    r = subprocess.run([tool, "seq.yuv", str(w), str(h), "22"], cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 1 and "model_2000000_qp20~25.dat" in r.stderr
    r = subprocess.run([tool, "seq.yuv", str(w + 8), str(h), str(qp)], cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr


def test_get_prob_frame_sub_range(pkg, oracle, tmp_path):
    """get_prob(n_frames_start, n_frames_end) (video_to_cu_depth.py:75-87: the first n_frames_start frames are read and dropped):
    the host mirror routes a sub-range to ethcnn_predict_yuv_range; the output file holds exactly those frames, bit-exact vs
    the oracle; an empty range gives an empty file; a range outside the file is an error."""
    v = pkg.video_to_cu_depth
    w, h, frames, qp = 416, 240, 7, 32
    yuv = _yuv(str(tmp_path / "seq.yuv"), w, h, frames, 5)
    blob = oracle.synth_blob(4, 8.0)
    c = pkg.EthCnn(device=0)
    c.load_blob(blob)
    c.set_thresholds(0.5, 0.5)
    out = str(tmp_path / "part.dat")
    luma_all = oracle.predict_frames(blob, yuv, w, h, frames, qp, 0.5, 0.5, frame_stride=w * h * 3 // 2).reshape(frames, -1, 21)
    for a, b in ((2, 5), (0, 7), (6, 7), (3, 3)):
        n = v.get_prob(c, str(tmp_path / "seq.yuv"), 64, out, qp, a, b, w, h)
        got = np.fromfile(out, dtype="<f4").reshape(-1, 21)
        assert n == b - a and got.shape[0] == (b - a) * 28
        assert np.array_equal(got.view(np.uint32), luma_all[a:b].reshape(-1, 21).view(np.uint32)), (a, b)
    for a, b in ((-1, 3), (2, 8), (5, 4)):
        with pytest.raises((ValueError, pkg.EthCnnError)):
            v.get_prob(c, str(tmp_path / "seq.yuv"), 64, out, qp, a, b, w, h)
    assert not [f for f in os.listdir(str(tmp_path)) if ".tmp." in f]
    c.close()

"""-m gpu: the HIP path against what THE REFERENCE'S OWN PYTHON FILES wrote (tests/golden/ref_exec_golden.npz, produced
in the build container by running video_to_cu_depth.py / resi_to_cu_depth_LDP.py over tests/tf_shim.py -- see
tests/test_ref_exec.py for what that does and does not pin; ref_exec_golden_torch.npz = the same runs with PyTorch's own
fp32 CPU kernels inside the tf.* calls, checked beside it at 3e-5).  Nothing here reads /root/reference.

AI: the drop-in command line in a directory holding Thr_info.txt and the four model bundles, exactly the reference's
file contract -> cu_depth.dat.  LDP: ethcnn_ldp_step over the recurrence (state resident in HBM), the reference's real
qp32 LSTM bundle, and the Python + native daemons over the file protocol.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from test_ref_exec import AI_TAGS, GOLDEN, GOLDEN_TORCH, TOL, TOL_TORCH, ai_case, gen, ldp_inputs, thr13
from tfckpt_writer import write_bundle

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LAUNCHER = os.path.join(ROOT, "video_to_cu_depth.py")
NATIVE = os.path.join(ROOT, "hevc-complexity-reduction_amd", "bin", "video_to_cu_depth")
AI_MODEL_NAMES = {22: "model_2000000_qp20~25.dat", 27: "model_2000000_qp25~30.dat",
                  32: "model_2000000_qp30~35.dat", 37: "model_2000000_qp35~40.dat"}


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLDEN)


@pytest.fixture(scope="module")
def gold_torch():
    return np.load(GOLDEN_TORCH)


@pytest.fixture(scope="module")
def model_dir(pkg, oracle, tmp_path_factory):
    """the four bundles video_to_cu_depth.py:126-133 chooses between, one seeded blob per band"""
    d = tmp_path_factory.mktemp("models")
    for band, blob in gen.blobs().items():
        write_bundle(str(d / AI_MODEL_NAMES[band]), [(n, np.array(v)) for n, v in oracle.tensor_views(blob).items()],
                     data_crc=pkg.ethcnn.crc32c_masked)
    return d


def _workdir(tmp_path, model_dir, thr_text, luma_frames):
    import ctu_gen
    for f in os.listdir(str(model_dir)):
        os.symlink(os.path.join(str(model_dir), f), str(tmp_path / f))
    (tmp_path / "Thr_info.txt").write_text(thr_text)
    (tmp_path / "in.yuv").write_bytes(ctu_gen.yuv420_bytes(list(luma_frames)))


@pytest.mark.parametrize("tag", AI_TAGS)
def test_drop_in_command_line_matches_the_reference_scripts_output(gold, gold_torch, oracle, model_dir, tmp_path, tag):
    w, h, nf, qp, luma, blob = ai_case(gold, tag)
    _workdir(tmp_path, model_dir, str(gold[tag + "_thr"]), luma)
    # the Python launcher HM's unchanged hook runs; for the big frames also the C99 tool over the same ABI
    cmds = [[sys.executable, LAUNCHER]] + ([[NATIVE]] if tag.startswith("ai_big") or tag == "ai_small" else [])
    want = gold[tag + "_probs"]
    for cmd in cmds:
        if os.path.exists(str(tmp_path / "cu_depth.dat")):
            os.remove(str(tmp_path / "cu_depth.dat"))
        r = subprocess.run(cmd + ["in.yuv", str(w), str(h), str(qp)], cwd=str(tmp_path), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        got = np.fromfile(str(tmp_path / "cu_depth.dat"), dtype="<f4").reshape(-1, 21)
        assert got.shape == want.shape
        assert np.array_equal(got == 0, want == 0), "gate pattern differs from the reference run"
        assert np.abs(got - want).max() <= TOL
        # the same run of the reference's script with torch's fp32 kernels behind the tf.* calls
        assert str(gold_torch[tag + "_thr"]) == str(gold[tag + "_thr"])
        assert np.array_equal(got == 0, gold_torch[tag + "_probs"] == 0) and np.abs(got - gold_torch[tag + "_probs"]).max() <= TOL_TORCH
        # and the usual bar: bit-exact against the oracle on the same inputs
        t1, t2 = thr13(gold, tag)
        ora = oracle.predict_frames(blob, luma, w, h, nf, qp, t1, t2)
        assert np.array_equal(got.view(np.uint32), ora.view(np.uint32))


@pytest.mark.parametrize("tag", ["ldp_a", "ldp_b"])
def test_ldp_step_matches_the_reference_daemon(pkg, gold, gold_torch, tag):
    w, h, qp, cnn, lstm, frames, i_frames = ldp_inputs(gold)
    t1, t2 = thr13(gold, tag)
    with pkg.EthCnn(device=0) as c:
        c.load_blob(cnn)
        c.load_lstm_checkpoint(os.path.join(HERE, "golden", "model_LDP_200000_qp32.dat"))
        c.set_thresholds(t1, t2)
        for k, (luma, i_frame) in enumerate(zip(frames, i_frames)):
            P = c.ldp_step(luma, w, h, qp, i_frame)
            want = gold[tag + "_probs"][k]
            assert np.array_equal(P == 0, want == 0), (tag, i_frame)
            assert np.abs(P - want).max() <= TOL, (tag, i_frame)
            wt = gold_torch[tag + "_probs"][k]
            assert np.array_equal(P == 0, wt == 0) and np.abs(P - wt).max() <= TOL_TORCH, (tag, i_frame)
            if tag == "ldp_a":
                S = c.ldp_get_state(w, h)
                assert np.abs(S.reshape(-1) - gold["ldp_a_state"][k].reshape(-1)).max() <= TOL, i_frame
                assert np.abs(S.reshape(-1) - gold_torch["ldp_a_state"][k].reshape(-1)).max() <= TOL_TORCH, i_frame


@pytest.mark.parametrize("native", [False, True])
def test_ldp_daemons_match_the_reference_daemon_over_the_file_protocol(pkg, oracle, gold, tmp_path, native):
    """the same handshake the fixture was produced with (tests/ref_exec.py::LdpDaemon.frame = TEncGOP.cpp:1471-1497),
    against the drop-in daemons: cu_depth.dat and state.dat of every frame"""
    import time
    w, h, qp, cnn, lstm, frames, i_frames = ldp_inputs(gold)
    d = tmp_path
    (d / "Thr_info.txt").write_text(str(gold["ldp_a_thr"]))
    write_bundle(str(d / "model_LDP_2000000_qp22~37.dat"), [(n, np.array(v)) for n, v in oracle.tensor_views(cnn).items()],
                 data_crc=pkg.ethcnn.crc32c_masked)
    for ext in (".index", ".data-00000-of-00001"):
        os.symlink(os.path.join(HERE, "golden", "model_LDP_200000_qp32.dat" + ext), str(d / ("model_LDP_200000_qp32.dat" + ext)))
    cmd = [sys.executable, os.path.join(ROOT, "resi_to_cu_depth_LDP.py"), "--native" if native else "--python",
           "--max-frames", str(len(frames)), "--idle-timeout", "60"]
    log = open(str(d / "daemon.log"), "wb")
    proc = subprocess.Popen(cmd, cwd=str(d), stdout=log, stderr=subprocess.STDOUT)
    try:
        for k, (luma, i_frame) in enumerate(zip(frames, i_frames)):
            (d / "resi.yuv").write_bytes(luma.tobytes() + bytes([128]) * (w * h // 2))
            if (d / "pred_end.sig").exists():
                os.remove(str(d / "pred_end.sig"))
            with open(str(d / "command.dat"), "w+") as f:
                f.write("%d %d %d %d [end]" % (i_frame, w, h, qp))
            open(str(d / "pred_start.sig"), "w+").close()
            t0 = time.time()
            while not (d / "pred_end.sig").exists():
                assert proc.poll() is None, open(str(d / "daemon.log"), errors="replace").read()[-2000:]
                assert time.time() - t0 < 120
                time.sleep(0.002)
            os.remove(str(d / "pred_end.sig"))
            P = np.fromfile(str(d / "cu_depth.dat"), dtype="<f4").reshape(-1, 21)
            want = gold["ldp_a_probs"][k]
            assert np.array_equal(P == 0, want == 0) and np.abs(P - want).max() <= TOL, i_frame
            # state.dat is refreshed behind the ending signal (see the daemon's header): wait for the sidecar to name this frame
            t0 = time.time()
            while True:
                try:
                    if open(str(d / "state.dat.idx")).read().split()[:1] == [str(i_frame)]:
                        break
                except (IOError, OSError):
                    pass
                assert time.time() - t0 < 30
                time.sleep(0.002)
            S = np.fromfile(str(d / "state.dat"), dtype="<f4")
            assert np.abs(S - gold["ldp_a_state"][k].reshape(-1)).max() <= TOL, i_frame
        proc.wait(timeout=60)
        assert proc.returncode == 0
    finally:
        if proc.poll() is None:
            proc.kill()
            proc.wait()
        log.close()

"""-m gpu: the native Low-Delay-P daemon (tools/resi_to_cu_depth_ldp.c, C over the ABI, inotify wake-up) against the Python
daemon (hevc-complexity-reduction_amd/resi_to_cu_depth_LDP.py), both driven by tools/ldp_client.c = HM's side of the file
handshake (TEncGOP.cpp:1466-1506): every frame's cu_depth.dat byte-identical (per-frame digests), the state.dat left behind
byte-identical, the sidecar in its final form; a stale sidecar stops the native daemon with a message that names the recovery,
--accept-stale takes the file as it is."""
import hashlib
import os
import shutil
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "hevc-complexity-reduction_amd", "bin")
GOLD = os.path.join(ROOT, "tests", "golden", "model_LDP_200000_qp32.dat")


def _workdir(path):
    os.makedirs(path)
    open(os.path.join(path, "Thr_info.txt"), "w").write("0.4 0.6 0.3 0.7 0.2 0.8")
    for ext in (".index", ".data-00000-of-00001"):
        shutil.copy(GOLD + ext, os.path.join(path, "model_LDP_200000_qp32.dat" + ext))
    return path


def _serve(kind, work, frames, extra=()):
    env = dict(os.environ, ETHCNN_SYNTHETIC_SEED="21")
    if kind == "python":
        os.symlink(os.path.join(ROOT, "resi_to_cu_depth_LDP.py"), os.path.join(work, "resi_to_cu_depth_LDP.py"))
        cmd = [sys.executable, "resi_to_cu_depth_LDP.py", "--max-frames", str(frames), "--idle-timeout", "60"]
        if not set(extra) & {"--native", "--default"}:
            cmd.append("--python")  # the launcher's default is the C daemon (when built); the Python one is the opt-out
        extra = [a for a in extra if a != "--default"]
    else:
        cmd = [os.path.join(BIN, "resi_to_cu_depth_ldp"), "--max-frames", str(frames), "--idle-timeout", "60", "--quiet"]
    d = subprocess.Popen(cmd + list(extra), cwd=work, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    time.sleep(6.0 if kind == "python" else 2.5)
    return d


def _client(work, w, h, frames, seed=7):
    return subprocess.run([os.path.join(BIN, "ldp_client"), work, str(w), str(h), "32", str(frames), "--seed", str(seed), "--digest",
                           os.path.join(work, "digest.txt")], capture_output=True, text=True, timeout=300)


@pytest.mark.parametrize("w,h,frames", [(1920, 1080, 40), (416, 240, 60), (200, 136, 12)])
def test_native_daemon_matches_the_python_daemon(tmp_path, w, h, frames):
    out = {}
    # native: resi.yuv streamed into the running prediction (pictures of >= 512 KiB: ethcnn_ldp_step_begin / rows_ready / end, eight
    # reader threads); native-plain: --no-stream --spin (read first, then ethcnn_ldp_step; the reference daemon's busy wait)
    for kind in ("python", "native", "native-plain"):
        work = _workdir(str(tmp_path / kind))
        d = _serve(kind, work, frames, extra=("--no-stream", "--spin") if kind == "native-plain" else ())
        c = _client(work, w, h, frames)
        d.wait(timeout=60)
        assert c.returncode == 0 and d.returncode == 0, (kind, c.stderr[-500:], d.stderr.read()[-800:])
        assert "handshake p50" in c.stdout
        out[kind] = (open(os.path.join(work, "digest.txt")).read(), hashlib.md5(open(os.path.join(work, "state.dat"), "rb").read()).hexdigest(),
                     open(os.path.join(work, "state.dat.idx")).read().split())
        assert not [f for f in os.listdir(work) if ".tmp." in f]
    assert out["python"][0] == out["native"][0] == out["native-plain"][0], "per-frame cu_depth.dat digests differ"
    assert out["python"][1] == out["native-plain"][1]
    assert len(out["native"][0].splitlines()) == frames
    assert out["python"][1] == out["native"][1], "state.dat differs"
    assert out["python"][2] == out["native"][2] == [str(frames), str(w), str(h)]


def test_launcher_native_switch(tmp_path):
    """`python resi_to_cu_depth_LDP.py` (the drop-in launcher, no flag) serves the handshake with the C daemon -- so does --native;
    --python keeps the Python daemon: same answers from all three."""
    out = {}
    for kind in ("python", "launcher-native", "launcher-default"):
        work = _workdir(str(tmp_path / kind))
        d = _serve("python", work, 6, extra={"python": (), "launcher-native": ("--native", "--quiet"), "launcher-default": ("--default", "--quiet")}[kind])
        c = _client(work, 416, 240, 6)
        d.wait(timeout=60)
        assert c.returncode == 0 and d.returncode == 0, (kind, c.stderr[-500:], d.stderr.read()[-800:])
        out[kind] = open(os.path.join(work, "digest.txt")).read()
    assert out["python"] == out["launcher-native"] == out["launcher-default"] and len(out["python"].splitlines()) == 6


def test_native_daemon_restart_and_stale_state(tmp_path):
    """A second daemon continues a sequence from state.dat (frames 4.. after a restart = the same digests as one daemon serving
    all of them); a sidecar left at "pending" stops it (exit 1, message with the recovery step) unless --accept-stale."""
    w, h = 416, 240
    ref = _workdir(str(tmp_path / "one"))
    d = _serve("native", ref, 6)
    assert _client(ref, w, h, 6).returncode == 0
    d.wait(timeout=60)
    want = open(os.path.join(ref, "digest.txt")).read().splitlines()
    # the same six frames, daemon restarted after three: ldp_client numbers frames from 1, so the second half is replayed with
    # an offset client: run 6 frames against daemon A (3 frames) + daemon B (3 frames)
    two = _workdir(str(tmp_path / "two"))
    a = _serve("native", two, 3)
    cl = subprocess.Popen([os.path.join(BIN, "ldp_client"), two, str(w), str(h), "32", "6", "--seed", "7", "--digest", os.path.join(two, "digest.txt")],
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    a.wait(timeout=60)
    b = _serve("native", two, 3)  # starts while the client already waits for frame 4: state comes from state.dat + sidecar
    assert cl.wait(timeout=120) == 0, cl.stderr.read()[-500:]
    b.wait(timeout=60)
    assert open(os.path.join(two, "digest.txt")).read().splitlines() == want
    # stale sidecar
    open(os.path.join(two, "state.dat.idx"), "w").write("pending 6 %d %d\n" % (w, h))
    cl = subprocess.Popen([os.path.join(BIN, "ldp_client"), two, str(w), str(h), "32", "1", "--seed", "9"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    time.sleep(0.3)
    cl.kill()  # (it only had to leave command.dat for POC 1 ... which needs no state; ask for a later frame by hand instead)
    open(os.path.join(two, "command.dat"), "w").write("7 %d %d 32 [end]" % (w, h))
    open(os.path.join(two, "pred_start.sig"), "w").close()
    s = _serve("native", two, 1)
    assert s.wait(timeout=60) == 1
    msg = s.stderr.read()
    assert "stale" in msg and "state.dat.idx" in msg
    assert not os.path.exists(os.path.join(two, "pred_end.sig"))
    open(os.path.join(two, "pred_start.sig"), "w").close()
    s = _serve("native", two, 1, extra=("--accept-stale",))
    assert s.wait(timeout=60) == 0
    assert os.path.exists(os.path.join(two, "pred_end.sig"))
    assert open(os.path.join(two, "state.dat.idx")).read().split() == ["7", str(w), str(h)]

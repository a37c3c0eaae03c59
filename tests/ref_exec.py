"""TEST INFRASTRUCTURE (build container only: needs /root/reference): runs the reference's OWN Python files
as separate processes over tests/tf_shim.py (a numpy stand-in for the TensorFlow calls they make).

  run_ai(...)      `python video_to_cu_depth.py <yuv> <w> <h> <qp>` exactly as TAppEncCfg.cpp:2317-2321 launches it,
                   in a scratch directory holding Thr_info.txt and the four model bundles -> the bytes of cu_depth.dat
  LdpDaemon(...)   `python resi_to_cu_depth_LDP.py` (its __main__ loop), driven over its file protocol the way
                   TEncGOP.cpp:1471-1497 drives it -> cu_depth.dat and state.dat per frame

The reference's files are executed from where they lie (sys.path[0] = their bin/ directory); nothing of them is
copied.  Model bundles: the trained CNN .data blobs are absent from the reference, so seeded synthetic
blobs are written as real TF-V2 bundles (tests/tfckpt_writer.py) under the names the scripts restore; the LSTM
bundle is the reference's real model_LDP_200000_qp32.dat (symlinked).
"""
import json
import os
import signal
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for _p in (HERE, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

REF_AI_BIN = "/root/reference/HM-16.5_Test_AI/bin"
REF_LDP_BIN = "/root/reference/HM-16.5_Test_LDP/bin"
AI_MODEL_NAMES = {22: "model_2000000_qp20~25.dat", 27: "model_2000000_qp25~30.dat",
                  32: "model_2000000_qp30~35.dat", 37: "model_2000000_qp35~40.dat"}  # one QP per band -> file
LDP_CNN_NAME = "model_LDP_2000000_qp22~37.dat"

# the child: shim installed as `tensorflow`, the script's own directory first on sys.path (as `python script.py`
# would have it), real argv, run as __main__.  On exit / SIGTERM it leaves which stand-in ops ran and what was restored.
_BOOT = r"""
import json, os, runpy, signal, sys
sys.path.insert(0, %(tests)r)
import tf_shim
tf_shim.install()
script = %(script)r
def _report(*a):
    json.dump({"ops": sorted(tf_shim.OPS_USED), "restored": tf_shim.RESTORED, "kernels": tf_shim.KERNELS}, open("shim_report.json", "w"))
    if a:
        os._exit(0)
signal.signal(signal.SIGTERM, _report)
sys.path.insert(0, os.path.dirname(script))
sys.argv = [script] + %(argv)r
try:
    runpy.run_path(script, run_name="__main__")
finally:
    _report()
"""


def _child(script, argv, cwd, kernels="numpy", **popen):
    code = _BOOT % {"tests": HERE, "script": script, "argv": [str(a) for a in argv]}
    # never leave .pyc files next to the reference's sources; kernels: tf_shim's op kernels (numpy restatements | torch's own)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", TF_SHIM_KERNELS=kernels, OMP_NUM_THREADS="4")
    return subprocess.Popen([sys.executable, "-c", code], cwd=cwd, env=env, **popen)


def write_cnn_bundle(prefix, blob):
    import ethcnn_np as oracle
    from tfckpt_writer import write_bundle
    write_bundle(prefix, [(n, np.array(v)) for n, v in oracle.tensor_views(np.asarray(blob, dtype=np.float32)).items()])


def run_ai(workdir, yuv_bytes, w, h, qp, thr_text, blobs_by_band, kernels="numpy"):
    """blobs_by_band: {22|27|32|37: float32 blob} -> bundles under the four names video_to_cu_depth.py:126-133 restores.
    Returns (cu_depth.dat as float32 [n,21], report dict, stdout text)."""
    os.makedirs(workdir, exist_ok=True)
    with open(os.path.join(workdir, "Thr_info.txt"), "w") as f:
        f.write(thr_text)
    for band, blob in blobs_by_band.items():
        write_cnn_bundle(os.path.join(workdir, AI_MODEL_NAMES[band]), blob)
    with open(os.path.join(workdir, "in.yuv"), "wb") as f:
        f.write(yuv_bytes)
    for stale in ("cu_depth.dat", "shim_report.json"):
        if os.path.exists(os.path.join(workdir, stale)):
            os.remove(os.path.join(workdir, stale))
    p = _child(os.path.join(REF_AI_BIN, "video_to_cu_depth.py"), ["in.yuv", w, h, qp], workdir, kernels=kernels,
               stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out, _ = p.communicate(timeout=1800)
    text = out.decode(errors="replace")
    if p.returncode != 0:
        raise RuntimeError("reference video_to_cu_depth.py failed (rc %d):\n%s" % (p.returncode, text[-3000:]))
    probs = np.fromfile(os.path.join(workdir, "cu_depth.dat"), dtype="<f4").reshape(-1, 21)
    return probs, json.load(open(os.path.join(workdir, "shim_report.json"))), text


class LdpDaemon(object):
    """the reference's resi_to_cu_depth_LDP.py daemon in `workdir`, driven as HM-LDP drives it"""

    def __init__(self, workdir, thr_text, cnn_blob, lstm_prefixes, kernels="numpy"):
        """lstm_prefixes: {file name the daemon restores: existing bundle prefix} (symlinked into workdir)"""
        self.dir = workdir
        os.makedirs(workdir, exist_ok=True)
        with open(self._p("Thr_info.txt"), "w") as f:
            f.write(thr_text)
        write_cnn_bundle(self._p(LDP_CNN_NAME), cnn_blob)
        for name, src in lstm_prefixes.items():
            for ext in (".index", ".data-00000-of-00001"):
                if os.path.lexists(self._p(name + ext)):
                    os.remove(self._p(name + ext))
                os.symlink(src + ext, self._p(name + ext))
        for stale in ("pred_start.sig", "pred_end.sig", "state.dat", "cu_depth.dat", "command.dat", "shim_report.json"):
            if os.path.exists(self._p(stale)):
                os.remove(self._p(stale))
        self.log = open(self._p("daemon.log"), "wb")
        self.proc = _child(os.path.join(REF_LDP_BIN, "resi_to_cu_depth_LDP.py"), [], workdir, kernels=kernels,
                           stdout=self.log, stderr=subprocess.STDOUT)

    def _p(self, name):
        return os.path.join(self.dir, name)

    def frame(self, luma, i_frame, qp, timeout=600.0):
        """one handshake (TEncGOP.cpp:1471-1497) -> (cu_depth.dat [n,21], state.dat [n,1,2,448])"""
        h, w = luma.shape
        with open(self._p("resi.yuv"), "wb") as f:
            f.write(luma.tobytes() + bytes([128]) * (w * h // 2))
        if os.path.exists(self._p("pred_end.sig")):
            os.remove(self._p("pred_end.sig"))
        with open(self._p("command.dat"), "w+") as f:
            f.write("%d %d %d %d [end]" % (i_frame, w, h, qp))
        open(self._p("pred_start.sig"), "w+").close()
        t0 = time.time()
        while not os.path.exists(self._p("pred_end.sig")):
            if self.proc.poll() is not None:
                raise RuntimeError("reference LDP daemon died:\n" + open(self._p("daemon.log"), errors="replace").read()[-3000:])
            if time.time() - t0 > timeout:
                raise RuntimeError("reference LDP daemon: no pred_end.sig within %.0f s" % timeout)
            time.sleep(0.005)
        os.remove(self._p("pred_end.sig"))
        n = ((w + 63) // 64) * ((h + 63) // 64)
        probs = np.fromfile(self._p("cu_depth.dat"), dtype="<f4").reshape(n, 21)
        state = np.fromfile(self._p("state.dat"), dtype="<f4").reshape(n, 1, 2, 448)
        return probs, state

    def close(self):
        """stop exactly the process started here; -> report dict"""
        if self.proc.poll() is None:
            self.proc.send_signal(signal.SIGTERM)
            try:
                self.proc.wait(timeout=30)
            except subprocess.TimeoutExpired:
                self.proc.kill()
                self.proc.wait()
        self.log.close()
        rep = self._p("shim_report.json")
        return json.load(open(rep)) if os.path.exists(rep) else {}

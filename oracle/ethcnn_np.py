"""Python face of the CPU oracle (TEST INFRASTRUCTURE ONLY -- see oracle/ethcnn_oracle.c).

Holds three things:
  * ctypes bindings to oracle/_build/libethcnn_oracle.so (the C restatement, canonical and
    literal summation orders);
  * `forward64`: an INDEPENDENT numpy float64 restatement of the reference graph written in
    whole-branch form (reshape/matmul convs, no 21-unit decomposition), following
    /root/reference/HM-16.5_Test_AI/bin/net_CNN.py:103-187 -- used to pin the C oracle;
  * the checkpoint tensor table and the seeded synthetic-weight generator (the trained
    blobs are absent from the reference).

Nothing here is imported by the product package.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libethcnn_oracle.so")

NFEAT, NH1, NOUT = 2688, 448, 21

# (name, shape, byte offset) -- TF-V2 bundle order (keys sorted), SURVEY.md Appendix A.4,
# decoded from /root/reference/HM-16.5_Test_AI/bin/model_2000000_qp30~35.dat.index.
TENSORS = [
    ("Variable", (4, 4, 1, 16), 0), ("Variable_1", (16,), 1024),
    ("Variable_10", (2, 2, 24, 32), 1088), ("Variable_11", (32,), 13376),
    ("Variable_12", (4, 4, 1, 16), 13504), ("Variable_13", (16,), 14528),
    ("Variable_14", (2, 2, 16, 24), 14592), ("Variable_15", (24,), 20736),
    ("Variable_16", (2, 2, 24, 32), 20832), ("Variable_17", (32,), 33120),
    ("Variable_2", (2, 2, 16, 24), 33248), ("Variable_3", (24,), 39392),
    ("Variable_4", (2, 2, 24, 32), 39488), ("Variable_5", (32,), 51776),
    ("Variable_6", (4, 4, 1, 16), 51904), ("Variable_7", (16,), 52928),
    ("Variable_8", (2, 2, 16, 24), 52992), ("Variable_9", (24,), 59136),
    ("h_fc1__16__b", (256,), 59232), ("h_fc1__16__w", (2688, 256), 60256),
    ("h_fc1__32__b", (128,), 2812768), ("h_fc1__32__w", (2688, 128), 2813280),
    ("h_fc1__64__b", (64,), 4189536), ("h_fc1__64__w", (2688, 64), 4189792),
    ("h_fc2__16__b", (192,), 4877920), ("h_fc2__16__w", (257, 192), 4878688),
    ("h_fc2__32__b", (96,), 5076064), ("h_fc2__32__w", (129, 96), 5076448),
    ("h_fc2__64__b", (48,), 5125984), ("h_fc2__64__w", (65, 48), 5126176),
    ("y_conv_flat__16__b", (16,), 5138656), ("y_conv_flat__16__w", (193, 16), 5138720),
    ("y_conv_flat__32__b", (4,), 5151072), ("y_conv_flat__32__w", (97, 4), 5151088),
    ("y_conv_flat__64__b", (1,), 5152640), ("y_conv_flat__64__w", (49, 1), 5152644),
]
BLOB_BYTES = 5152840
BLOB_FLOATS = BLOB_BYTES // 4
# conv variable numbering: net_CNN.py:126-141 creates L, then M, then S.
BRANCH_VAR_BASE = {"L": 0, "M": 6, "S": 12}


def tensor_views(blob):
    """dict name -> ndarray view (reshaped) into a flat float32 blob."""
    out = {}
    for name, shape, off in TENSORS:
        n = int(np.prod(shape))
        out[name] = blob[off // 4: off // 4 + n].reshape(shape)
    return out


def _var(i):
    return "Variable" if i == 0 else "Variable_%d" % i


# ----------------------------------------------------------------- synthetic weights ---
_M64 = (1 << 64) - 1


def _splitmix64(z):
    """One splitmix64 output for state z (numpy uint64 array or python int)."""
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def synth_blob(seed=1, head_gain=1.0):
    """Counter-based synthetic model blob (float32[BLOB_FLOATS]) in checkpoint layout.

    element i of tensor t:  key = splitmix64(seed ^ (0xD6E8FEB86659FD93 * (t+1)))
                            h   = splitmix64(key + i);  u = h >> 40  (24 bits)
                            val = ((u + 0.5) / 2^23 - 1) * scale      (float64, then -> f32)
    scale = sqrt(3/fan_in) for weights (x head_gain for h_fc2* / y_conv_flat* weights),
            0.1 for biases.  Same generator in csrc/ethcnn_weights.cpp (ethcnn_load_synthetic).
    """
    blob = np.zeros(BLOB_FLOATS, dtype=np.float32)
    for t, (name, shape, off) in enumerate(TENSORS):
        n = int(np.prod(shape))
        key = _splitmix64((int(seed) ^ ((0xD6E8FEB86659FD93 * (t + 1)) & _M64)) & _M64)
        with np.errstate(over="ignore"):
            h = _splitmix64(key + np.arange(n, dtype=np.uint64))
        u = (h >> np.uint64(40)).astype(np.float64)
        val = (u + 0.5) * (1.0 / 8388608.0) - 1.0
        if len(shape) == 1:
            scale = 0.1
        else:
            scale = np.sqrt(3.0 / float(np.prod(shape[:-1])))
            if name.startswith("h_fc2") or name.startswith("y_conv"):
                scale = scale * float(head_gain)
        blob[off // 4: off // 4 + n] = (val * scale).astype(np.float32)
    return blob


# ------------------------------------------------------------------- C oracle binding ---
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _usable_cpus():
    """logical CPUs of this process, capped by the cgroup CPU quota (the GPU boxes: 256 logical, quota 16 cores;
    256 OpenMP threads under that quota run 5x SLOWER than 16)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per) + 0.5)))
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            if q > 0:
                n = max(1, min(n, int(q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()) + 0.5)))
        except Exception:
            pass
    return n


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        fp = ctypes.POINTER(ctypes.c_float)
        up = ctypes.POINTER(ctypes.c_uint8)
        L.oracle_features.argtypes = [fp, up, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp]
        L.oracle_fc1.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, fp]
        L.oracle_heads.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, fp]
        L.oracle_gates.argtypes = [fp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float]
        L.oracle_tile_frame.argtypes = [up, ctypes.c_int, ctypes.c_int, ctypes.c_long, up]
        L.oracle_predict_frames.argtypes = [fp, up, ctypes.c_int, ctypes.c_int, ctypes.c_long,
                                            ctypes.c_long, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_float, ctypes.c_float, ctypes.c_int, fp]
        L.oracle_resi_vectors.argtypes = [fp, up, ctypes.c_int, ctypes.c_int, ctypes.c_long,
                                          ctypes.c_int, fp]
        L.oracle_set_threads.argtypes = [ctypes.c_int]
        L.oracle_set_threads(_usable_cpus())  # never more OpenMP threads than the cgroup lets run (CFS throttling otherwise)
        L.oracle_expf_export.argtypes = [ctypes.c_float]
        L.oracle_expf_export.restype = ctypes.c_float
        _lib = L
    return _lib


def _f(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _u(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def set_threads(n):
    """OpenMP team size of the C oracle; returns the previous maximum."""
    return int(lib().oracle_set_threads(int(n)))


def _blob(blob):
    blob = np.ascontiguousarray(blob, dtype=np.float32)
    assert blob.size == BLOB_FLOATS
    return blob


def features(blob, ctus, mode=0, resi=0):
    blob = _blob(blob)
    ctus = np.ascontiguousarray(ctus, dtype=np.uint8).reshape(-1, 64, 64)
    F = np.empty((ctus.shape[0], NFEAT), dtype=np.float32)
    lib().oracle_features(_f(blob), _u(ctus), ctus.shape[0], mode, resi, _f(F))
    return F


def fc1(blob, F, mode=0):
    blob = _blob(blob)
    F = np.ascontiguousarray(F, dtype=np.float32)
    H1 = np.empty((F.shape[0], NH1), dtype=np.float32)
    lib().oracle_fc1(_f(blob), _f(F), F.shape[0], mode, _f(H1))
    return H1


def heads(blob, H1, qp, mode=0):
    """-> (ungated probs [n,21], logits [n,21])"""
    blob = _blob(blob)
    H1 = np.ascontiguousarray(H1, dtype=np.float32)
    P = np.empty((H1.shape[0], NOUT), dtype=np.float32)
    Z = np.empty((H1.shape[0], NOUT), dtype=np.float32)
    lib().oracle_heads(_f(blob), _f(H1), H1.shape[0], int(qp), mode, _f(P), _f(Z))
    return P, Z


def gates(probs, thr1, thr2, chunk=1024):
    P = np.array(probs, dtype=np.float32, order="C", copy=True)
    lib().oracle_gates(_f(P), P.shape[0], chunk, thr1, thr2)
    return P


def tile_frame(luma, w, h, pitch=None):
    luma = np.ascontiguousarray(luma, dtype=np.uint8)
    pitch = w if pitch is None else pitch
    n = ((w + 63) // 64) * ((h + 63) // 64)
    out = np.empty((n, 64, 64), dtype=np.uint8)
    got = lib().oracle_tile_frame(_u(luma), w, h, pitch, _u(out))
    assert got == n
    return out


def predict_frames(blob, luma, w, h, nframes, qp, thr1=0.5, thr2=0.5, mode=0, pitch=None,
                   frame_stride=None):
    """luma: uint8 buffer holding nframes planes (frame_stride bytes apart) -> [nframes*nctu, 21]"""
    blob = _blob(blob)
    luma = np.ascontiguousarray(luma, dtype=np.uint8)
    pitch = w if pitch is None else pitch
    frame_stride = pitch * h if frame_stride is None else frame_stride
    nctu = ((w + 63) // 64) * ((h + 63) // 64)
    P = np.empty((nframes * nctu, NOUT), dtype=np.float32)
    rc = lib().oracle_predict_frames(_f(blob), _u(luma), w, h, pitch, frame_stride, nframes, int(qp),
                                     thr1, thr2, mode, _f(P))
    assert rc == 0
    return P


def resi_vectors(blob, luma, w, h, mode=0, pitch=None):
    blob = _blob(blob)
    luma = np.ascontiguousarray(luma, dtype=np.uint8)
    pitch = w if pitch is None else pitch
    nctu = ((w + 63) // 64) * ((h + 63) // 64)
    V = np.empty((nctu, NH1), dtype=np.float32)
    rc = lib().oracle_resi_vectors(_f(blob), _u(luma), w, h, pitch, mode, _f(V))
    assert rc == 0
    return V


# -------------------------------------------- independent float64 restatement (numpy) ---
def _lrelu(x):
    return np.maximum(0.2 * x, x)  # alpha: float64(0.2); the f32 alpha differs by 3e-9


def _conv_nonoverlap(x, W, b, k):
    """x [n,H,W,C] float64; W HWIO [k,k,C,Co]; VALID, stride k (net_CNN.py:86-92)."""
    n, H, Wd, C = x.shape
    p = x.reshape(n, H // k, k, Wd // k, k, C).transpose(0, 1, 3, 2, 4, 5).reshape(n, H // k, Wd // k, k * k * C)
    return _lrelu(p @ W.reshape(k * k * C, -1).astype(np.float64) + b.astype(np.float64))


def forward64(blob, ctus, qp, resi=False):
    """float64 whole-graph restatement. Returns dict(F, H1, logits, probs) (ungated)."""
    tv = tensor_views(np.asarray(blob, dtype=np.float32))
    x = np.asarray(ctus, dtype=np.float64).reshape(-1, 64, 64)
    n = x.shape[0]
    if resi:
        x = (x - 128.0) / 255.0 * 10.0            # net_CNN_LSTM_one_step.py:153
    else:
        x = x * np.float64(np.float32(1.0 / 255.0))  # net_CNN.py:105 (f32 scalar constant)
    feats3, feats2 = {}, {}
    for br, pool in (("L", 4), ("M", 2), ("S", 1)):
        side = 64 // pool
        xb = x.reshape(n, side, pool, side, pool).mean(axis=(2, 4))               # aver_pool :62-63
        nb = side // 16
        m = xb.reshape(n, nb, 16, nb, 16).mean(axis=(2, 4), keepdims=True)        # :78-84
        xb = (xb.reshape(n, nb, 16, nb, 16) - m).reshape(n, side, side, 1)
        base = BRANCH_VAR_BASE[br]
        c1 = _conv_nonoverlap(xb, tv[_var(base)], tv[_var(base + 1)], 4)
        c2 = _conv_nonoverlap(c1, tv[_var(base + 2)], tv[_var(base + 3)], 2)
        c3 = _conv_nonoverlap(c2, tv[_var(base + 4)], tv[_var(base + 5)], 2)
        feats2[br], feats3[br] = c2.reshape(n, -1), c3.reshape(n, -1)
    F = np.concatenate([feats3["S"], feats3["M"], feats3["L"], feats2["S"], feats2["M"], feats2["L"]], axis=1)
    qn = float(qp) * np.float64(np.float32(1.0 / 51.0))                           # :106
    qcol = np.full((n, 1), qn)
    H1s, logits = [], []
    for tag in ("64", "32", "16"):
        h1 = _lrelu(F @ tv["h_fc1__%s__w" % tag].astype(np.float64) + tv["h_fc1__%s__b" % tag])
        H1s.append(h1)
        h2 = _lrelu(np.concatenate([h1, qcol], 1) @ tv["h_fc2__%s__w" % tag].astype(np.float64) + tv["h_fc2__%s__b" % tag])
        z = np.concatenate([h2, qcol], 1) @ tv["y_conv_flat__%s__w" % tag].astype(np.float64) + tv["y_conv_flat__%s__b" % tag]
        logits.append(z)
    Z = np.concatenate(logits, 1)
    return {"F": F, "H1": np.concatenate(H1s, 1), "logits": Z, "probs": 1.0 / (1.0 + np.exp(-Z))}

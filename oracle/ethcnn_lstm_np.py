"""Oracle face for the ETH-LSTM one-step row (TEST INFRASTRUCTURE ONLY).

ctypes binding of oracle_lstm_step (oracle/ethcnn_oracle.c), the LSTM checkpoint tensor table
(= /root/reference/HM-16.5_Test_LDP/bin/model_LDP_200000_qp*.dat.index), a seeded synthetic
blob generator, and an independent numpy float64 restatement of
net_CNN_LSTM_one_step.py:201-323 (LSTMCell with forget_bias 1, cell_clip 5; fc2/fc3 with efs).
"""
import ctypes

import numpy as np

import ethcnn_np as base

# (name, shape, byte offset) in bundle (sorted-key) order
LSTM_TENSORS = [
    ("RNN16/fc2/full_connect_b", (192,), 0), ("RNN16/fc2/full_connect_w", (261, 192), 768),
    ("RNN16/fc3/full_connect_b", (16,), 201216), ("RNN16/fc3/full_connect_w", (197, 16), 201280),
    ("RNN16/multi_rnn_cell/cell_0/lstm_cell/bias", (1024,), 213888),
    ("RNN16/multi_rnn_cell/cell_0/lstm_cell/kernel", (512, 1024), 217984),
    ("RNN32/fc2/full_connect_b", (96,), 2315136), ("RNN32/fc2/full_connect_w", (133, 96), 2315520),
    ("RNN32/fc3/full_connect_b", (4,), 2366592), ("RNN32/fc3/full_connect_w", (101, 4), 2366608),
    ("RNN32/multi_rnn_cell/cell_0/lstm_cell/bias", (512,), 2368224),
    ("RNN32/multi_rnn_cell/cell_0/lstm_cell/kernel", (256, 512), 2370272),
    ("RNN64/fc2/full_connect_b", (48,), 2894560), ("RNN64/fc2/full_connect_w", (69, 48), 2894752),
    ("RNN64/fc3/full_connect_b", (1,), 2908000), ("RNN64/fc3/full_connect_w", (53, 1), 2908004),
    ("RNN64/multi_rnn_cell/cell_0/lstm_cell/bias", (256,), 2908216),
    ("RNN64/multi_rnn_cell/cell_0/lstm_cell/kernel", (128, 256), 2909240),
]
LSTM_BLOB_BYTES = 3040312
LSTM_BLOB_FLOATS = LSTM_BLOB_BYTES // 4


def lstm_views(blob):
    return {n: blob[o // 4: o // 4 + int(np.prod(s))].reshape(s) for n, s, o in LSTM_TENSORS}


def synth_lstm_blob(seed=1, gain=1.0):
    """Same counter-based generator as ethcnn_np.synth_blob, tensor index t + 100; scale
    sqrt(3/fan_in) (x gain for the fc2 / fc3 matrices), biases 0.1."""
    blob = np.zeros(LSTM_BLOB_FLOATS, dtype=np.float32)
    for t, (name, shape, off) in enumerate(LSTM_TENSORS):
        n = int(np.prod(shape))
        key = base._splitmix64((int(seed) ^ ((0xD6E8FEB86659FD93 * (t + 101)) & base._M64)) & base._M64)
        with np.errstate(over="ignore"):
            h = base._splitmix64(key + np.arange(n, dtype=np.uint64))
        val = ((h >> np.uint64(40)).astype(np.float64) + 0.5) * (1.0 / 8388608.0) - 1.0
        if len(shape) == 1:
            scale = 0.1
        else:
            scale = np.sqrt(3.0 / float(shape[0]))
            if "/fc" in name:
                scale *= float(gain)
        blob[off // 4: off // 4 + n] = (val * scale).astype(np.float32)
    return blob


_bound = False


def _lib():
    global _bound
    L = base.lib()
    if not _bound:
        fp = ctypes.POINTER(ctypes.c_float)
        L.oracle_lstm_step.argtypes = [fp, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                       ctypes.c_float, ctypes.c_int, fp, fp]
        _bound = True
    return L


def lstm_step(lstm_blob, vec, state_in, qp, i_frame, thr1=0.5, thr2=0.5, mode=0):
    """vec [n,448], state_in [n,2,448] or None -> (probs [n,21] gated, state_out [n,2,448])"""
    lb = np.ascontiguousarray(lstm_blob, dtype=np.float32)
    assert lb.size == LSTM_BLOB_FLOATS
    vec = np.ascontiguousarray(vec, dtype=np.float32)
    n = vec.shape[0]
    P = np.empty((n, 21), dtype=np.float32)
    S = np.empty((n, 2, 448), dtype=np.float32)
    sin = None if state_in is None else np.ascontiguousarray(state_in, dtype=np.float32)
    rc = _lib().oracle_lstm_step(base._f(lb), base._f(vec), None if sin is None else base._f(sin), n, int(qp), int(i_frame),
                                 thr1, thr2, mode, base._f(P), base._f(S))
    assert rc == 0
    return P, S


def lstm_forward64(lstm_blob, vec, state_in, qp, i_frame):
    """float64 restatement: ungated probabilities + state_out."""
    tv = lstm_views(np.asarray(lstm_blob, dtype=np.float32))
    vec = np.asarray(vec, dtype=np.float64)
    n = vec.shape[0]
    st = np.zeros((n, 2, 448)) if state_in is None else np.asarray(state_in, dtype=np.float64)
    efs = np.zeros((n, 5))
    efs[:, 0] = float(qp) / 51.0 * 0.18
    efs[:, 1 + (i_frame % 4)] = 1.0
    sig = lambda x: 1.0 / (1.0 + np.exp(-x))
    lre = lambda x: np.maximum(0.2 * x, x)
    probs, so, o1 = [], np.zeros((n, 2, 448)), 0
    for tag, hid in (("64", 64), ("32", 128), ("16", 256)):
        x, c_prev, h_prev = vec[:, o1:o1 + hid], st[:, 0, o1:o1 + hid], st[:, 1, o1:o1 + hid]
        pre = "RNN%s/" % tag
        z = np.concatenate([x, h_prev], 1) @ tv[pre + "multi_rnn_cell/cell_0/lstm_cell/kernel"].astype(np.float64) \
            + tv[pre + "multi_rnn_cell/cell_0/lstm_cell/bias"]
        i, j, f, o = np.split(z, 4, axis=1)
        c = np.clip(sig(f + 1.0) * c_prev + sig(i) * np.tanh(j), -5.0, 5.0)
        h = sig(o) * np.tanh(c)
        so[:, 0, o1:o1 + hid], so[:, 1, o1:o1 + hid] = c, h
        h2 = lre(np.concatenate([h, efs], 1) @ tv[pre + "fc2/full_connect_w"].astype(np.float64) + tv[pre + "fc2/full_connect_b"])
        probs.append(sig(np.concatenate([h2, efs], 1) @ tv[pre + "fc3/full_connect_w"].astype(np.float64) + tv[pre + "fc3/full_connect_b"]))
        o1 += hid
    return np.concatenate(probs, 1), so

"""Independent numpy evaluation of the reference's batch gates (TEST INFRASTRUCTURE; no oracle, no HIP code involved).

Written from the two reference sites only:
  * net_CNN.py:175  y32 = tf.cond(count_nonzero(y64 > THR_L1_LOWER) > 0, y32_temp, zeros)
  * net_CNN.py:187  y16 = tf.cond(count_nonzero(y32 > THR_L2_LOWER) > 0, y16_temp, zeros)   (y32 = the GATED tensor)
  * video_to_cu_depth.py:61-73,88-109  the graph is fed sub-batches of <= 1024 consecutive CTUs of ONE frame
"""
import numpy as np


def gate_sub_batch(raw, thr1, thr2):
    """one sess.run: raw float32 [m <= 1024, 21] ungated [y64 | y32 | y16] -> gated copy (float32 compares, strict >)"""
    raw = np.asarray(raw, dtype=np.float32)
    y64, y32t, y16t = raw[:, 0:1], raw[:, 1:5], raw[:, 5:21]
    y32 = y32t if np.count_nonzero(y64 > np.float32(thr1)) > 0 else np.zeros_like(y32t)
    y16 = y16t if np.count_nonzero(y32 > np.float32(thr2)) > 0 else np.zeros_like(y16t)
    return np.concatenate([y64, y32, y16], axis=1)


def gate_frames(raw, nctu_per_frame, thr1, thr2, sub_batch_size=1024):
    """get_prob's frame loop around get_y_conv_on_large_data: sub-batches never cross a frame boundary"""
    raw = np.asarray(raw, dtype=np.float32)
    out = np.empty_like(raw)
    assert raw.shape[0] % nctu_per_frame == 0
    for f0 in range(0, raw.shape[0], nctu_per_frame):
        for s in range(f0, f0 + nctu_per_frame, sub_batch_size):
            e = min(s + sub_batch_size, f0 + nctu_per_frame)
            out[s:e] = gate_sub_batch(raw[s:e], thr1, thr2)
    return out

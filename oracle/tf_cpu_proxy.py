"""TF-CPU proxy ("B2" of BASELINE.md section 4) -- BASELINE INFRASTRUCTURE ONLY, never product code.

The reference's CPU path is Python + TensorFlow 1.x (video_to_cu_depth.py driving net_CNN.py).
TensorFlow is not installed here and cannot be, so the closest thing that can be TIMED on the GPU
box's host cores is this stand-in with the same COST STRUCTURE as the reference's timed scope
(`Predicting Time`, /root/reference/HM-16.5_Test_AI/bin/video_to_cu_depth.py:142-145):

  * one frame at a time: read w*h luma bytes + skip w*h/2 chroma bytes (:47-48), zero-pad to whole
    CTUs in a float64 array (:54-57);
  * a Python loop that copies every 64x64 CTU into a float64 [nctu,64,64,1] batch (:88-99), cast to
    float32 (:101);
  * the network in <= 1024-CTU feeds (:61-73), each feed = one framework call on all host threads
    (`sess.run` there, the torch-CPU graph below here: F.conv2d / avg_pool2d / matmul on NCHW
    tensors stand in for TF's Eigen kernels), batch gates per feed (net_CNN.py:175,187);
  * results collected in a float64 [frames*nctu,21] array, written once at the end (:78,114-116).

Its numbers are labelled "TF-CPU proxy" wherever they are printed.  Its OUTPUT is checked against the
C oracle (<= 1e-4, tests/test_cpu_proxy.py), so what is timed is the real computation.
"""
import math
import os
import time

import numpy as np


def _net(torch, F, tv, x_u8_f32, qp, thr1, thr2):
    """net() of net_CNN.py:103-187 for one fed sub-batch: x [n,64,64,1] float32 (raw 0..255)."""
    f32 = torch.float32
    n = x_u8_f32.shape[0]
    x = torch.from_numpy(x_u8_f32).reshape(n, 1, 64, 64) * torch.tensor(1.0 / 255.0, dtype=f32)
    qn = torch.full((n, 1), float(qp), dtype=f32) * torch.tensor(1.0 / 51.0, dtype=f32)
    alpha = torch.tensor(0.2, dtype=f32)

    def act(t):
        return torch.maximum(t * alpha, t)

    def conv(t, w, b, k):
        return act(F.conv2d(t, w, bias=None, stride=k) + b.view(1, -1, 1, 1))

    c2s, c3s = {}, {}
    for br, pool, side in (("L", 4, 16), ("M", 2, 32), ("S", 1, 64)):
        t = F.avg_pool2d(x, pool) if pool > 1 else x
        m = F.conv2d(t, tv["mean_k"], stride=16)
        t = t - F.interpolate(m, size=(side, side), mode="nearest")
        w = tv[br]
        c1 = conv(t, w[0], w[1], 4)
        c2 = conv(c1, w[2], w[3], 2)
        c3 = conv(c2, w[4], w[5], 2)
        c2s[br] = c2.permute(0, 2, 3, 1).reshape(n, -1)
        c3s[br] = c3.permute(0, 2, 3, 1).reshape(n, -1)
    feat = torch.cat([c3s["S"], c3s["M"], c3s["L"], c2s["S"], c2s["M"], c2s["L"]], 1)
    ys = []
    for tag in ("64", "32", "16"):
        h1 = act(feat @ tv["fc1w" + tag] + tv["fc1b" + tag])
        h2 = act(torch.cat([h1, qn], 1) @ tv["fc2w" + tag] + tv["fc2b" + tag])
        ys.append(torch.sigmoid(torch.cat([h2, qn], 1) @ tv["fc3w" + tag] + tv["fc3b" + tag]))
    y64, y32, y16 = ys
    if not bool((y64 > thr1).any()):
        y32 = torch.zeros_like(y32)
    if not bool((y32 > thr2).any()):
        y16 = torch.zeros_like(y16)
    return torch.cat([y64, y32, y16], 1).numpy()


def _weights(torch, blob):
    import ethcnn_np as o
    views = o.tensor_views(np.asarray(blob, dtype=np.float32))
    tv = {"mean_k": torch.full((1, 1, 16, 16), 1.0 / 256.0)}
    for br, base in (("L", 0), ("M", 6), ("S", 12)):
        ws = []
        for i in range(6):
            a = torch.from_numpy(np.array(views["Variable" if base + i == 0 else "Variable_%d" % (base + i)]))
            ws.append(a.permute(3, 2, 0, 1).contiguous() if a.dim() == 4 else a)  # HWIO -> OIHW
        tv[br] = ws
    for tag in ("64", "32", "16"):
        tv["fc1w" + tag] = torch.from_numpy(np.array(views["h_fc1__%s__w" % tag]))
        tv["fc1b" + tag] = torch.from_numpy(np.array(views["h_fc1__%s__b" % tag]))
        tv["fc2w" + tag] = torch.from_numpy(np.array(views["h_fc2__%s__w" % tag]))
        tv["fc2b" + tag] = torch.from_numpy(np.array(views["h_fc2__%s__b" % tag]))
        tv["fc3w" + tag] = torch.from_numpy(np.array(views["y_conv_flat__%s__w" % tag]))
        tv["fc3b" + tag] = torch.from_numpy(np.array(views["y_conv_flat__%s__b" % tag]))
    return tv


def _read_padded_luma(f, w, h):
    y = f.read(w * h)
    f.read(w * h // 2)
    if len(y) != w * h:
        raise IOError("short read")
    a = np.frombuffer(y, dtype=np.uint8).reshape(h, w)
    vh, vw = math.ceil(h / 64) * 64, math.ceil(w / 64) * 64
    if vh > h:
        a = np.concatenate((a, np.zeros((vh - h, w))), axis=0)   # promotes to float64, as the reference does
    if vw > w:
        a = np.concatenate((a, np.zeros((vh, vw - w))), axis=1)
    return a


def predict_file(blob, yuv_path, w, h, qp, out_path, thr1=0.5, thr2=0.5, max_frames=None, max_seconds=None,
                 threads=None):
    """The reference's timed scope on `yuv_path` -> `out_path`.  Stops early after `max_frames` frames
    or once `max_seconds` have elapsed (bounded bench sample).  Returns (frames_done, ctus_done, seconds)."""
    import torch
    import torch.nn.functional as F
    if threads:
        torch.set_num_threads(int(threads))
    tv = _weights(torch, blob)
    nframes = os.path.getsize(yuv_path) // (w * h * 3 // 2)
    if max_frames is not None:
        nframes = min(nframes, max_frames)
    cw, ch = math.ceil(w / 64), math.ceil(h / 64)
    nctu = cw * ch
    t0 = time.perf_counter()
    prob = np.zeros((nframes * nctu, 21))
    done = 0
    with torch.no_grad(), open(yuv_path, "rb") as f:
        for k in range(nframes):
            luma = _read_padded_luma(f, w, h)
            batch = np.zeros((nctu, 64, 64, 1))
            i = 0
            y0 = 0
            while y0 < h:
                x0 = 0
                while x0 < w:
                    batch[i] = luma[y0:y0 + 64, x0:x0 + 64].reshape(64, 64, 1)
                    i += 1
                    x0 += 64
                y0 += 64
            batch = batch.astype(np.float32)
            out = np.zeros((nctu, 21))
            for s in range(0, nctu, 1024):
                out[s:s + 1024] = _net(torch, F, tv, batch[s:s + 1024], qp, thr1, thr2)
            prob[k * nctu:(k + 1) * nctu] = out
            done = k + 1
            if max_seconds is not None and time.perf_counter() - t0 >= max_seconds:
                break
    with open(out_path, "wb") as fo:
        fo.write(prob[:done * nctu].astype(np.float32).tobytes())
    return done, done * nctu, time.perf_counter() - t0

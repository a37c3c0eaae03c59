#!/usr/bin/env python
"""Generates tests/golden/golden_v1.npz -- run HERE (build container), commit the output.

An INDEPENDENT PyTorch-CPU fp32 restatement of the reference graph
(/root/reference/HM-16.5_Test_AI/bin/net_CNN.py:103-187), written with torch's own conv /
pool / matmul kernels (F.conv2d on OIHW weights, F.avg_pool2d, nearest upsampling of the
block mean, NHWC flatten) -- no code shared with oracle/ or the product.  TensorFlow cannot
be imported in this container and the trained weights are absent, so the vectors use the
seeded synthetic weights; they pin the C oracle and (through it) the HIP kernels.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import ethcnn_np as o  # only for the tensor table + synthetic weight generator


def torch_forward(blob, ctus, qp, resi=False):
    tv = {k: torch.from_numpy(np.array(v)) for k, v in o.tensor_views(blob).items()}
    x = torch.from_numpy(ctus.astype(np.float32)).reshape(-1, 1, 64, 64)
    if resi:
        x = (x - 128) / 255.0 * 10                      # net_CNN_LSTM_one_step.py:153
    else:
        x = x * torch.tensor(1.0 / 255.0, dtype=torch.float32)   # net_CNN.py:105
    qn = torch.full((x.shape[0], 1), float(qp), dtype=torch.float32) * torch.tensor(1 / 51.0, dtype=torch.float32)

    def lrelu(t):
        return torch.maximum(t * torch.tensor(0.2, dtype=torch.float32), t)

    def conv(t, w_hwio, b, k):                       # non_overlap_conv :86-92
        w = w_hwio.permute(3, 2, 0, 1).contiguous()  # HWIO -> OIHW
        return lrelu(F.conv2d(t, w, bias=None, stride=k) + b.view(1, -1, 1, 1))

    def norm_local(t, width):                        # zero_mean_norm_local :78-84
        m = F.conv2d(t, torch.full((1, 1, 16, 16), 1.0 / 256.0), stride=16)
        return t - F.interpolate(m, size=(width, width), mode="nearest")

    feats2, feats3 = {}, {}
    for br, pool, width, base in (("L", 4, 16, 0), ("M", 2, 32, 6), ("S", 1, 64, 12)):
        t = F.avg_pool2d(x, pool) if pool > 1 else x
        t = norm_local(t, width)
        v = lambda i: tv["Variable" if base + i == 0 else "Variable_%d" % (base + i)]
        c1 = conv(t, v(0), v(1), 4)
        c2 = conv(c1, v(2), v(3), 2)
        c3 = conv(c2, v(4), v(5), 2)
        feats2[br] = c2.permute(0, 2, 3, 1).reshape(x.shape[0], -1)   # NHWC flatten
        feats3[br] = c3.permute(0, 2, 3, 1).reshape(x.shape[0], -1)
    feat = torch.cat([feats3["S"], feats3["M"], feats3["L"], feats2["S"], feats2["M"], feats2["L"]], 1)
    h1s, ys = [], []
    for tag in ("64", "32", "16"):
        h1 = lrelu(feat @ tv["h_fc1__%s__w" % tag] + tv["h_fc1__%s__b" % tag])
        h1s.append(h1)
        h2 = lrelu(torch.cat([h1, qn], 1) @ tv["h_fc2__%s__w" % tag] + tv["h_fc2__%s__b" % tag])
        ys.append(torch.sigmoid(torch.cat([h2, qn], 1) @ tv["y_conv_flat__%s__w" % tag] + tv["y_conv_flat__%s__b" % tag]))
    return feat.numpy(), torch.cat(h1s, 1).numpy(), torch.cat(ys, 1).numpy()


def make_ctus(seed, n):
    rng = np.random.default_rng(seed)
    ctus = rng.integers(0, 256, size=(n, 64, 64), dtype=np.uint8)
    yy, xx = np.mgrid[0:64, 0:64]
    k = n // 4
    ctus[:k] = ((yy * 2 + xx)[None] + rng.integers(0, 8, size=(k, 64, 64))).clip(0, 255).astype(np.uint8)
    ctus[k:2 * k] = rng.integers(0, 256, size=(k, 1, 1), dtype=np.uint8)
    ctus[2 * k] = 0
    ctus[2 * k + 1] = 255
    ctus[2 * k + 2, :, 40:] = 0      # a right-edge CTU after zero padding
    ctus[2 * k + 3, 24:, :] = 0      # a bottom-edge CTU
    return ctus


def main():
    torch.set_num_threads(1)
    out = {}
    n = 48
    ctus = make_ctus(2024, n)
    out["ctus"] = ctus
    for tag, seed, gain, qp in (("a", 1, 1.0, 32), ("b", 2, 8.0, 22)):
        blob = o.synth_blob(seed, gain)
        feat, h1, probs = torch_forward(blob, ctus, qp)
        out["%s_seed_gain_qp" % tag] = np.array([seed, gain, qp], dtype=np.float64)
        out["%s_probs" % tag] = probs.astype(np.float32)              # ungated [48,21]
        out["%s_feat4" % tag] = feat[:4].astype(np.float32)           # [4,2688]
        out["%s_h1_4" % tag] = h1[:4].astype(np.float32)              # [4,448]
    # config #5 front-end
    rng = np.random.default_rng(5)
    resi = np.clip(np.rint(128 + rng.laplace(0, 6, size=(8, 64, 64))), 0, 255).astype(np.uint8)
    blob = o.synth_blob(3, 1.0)
    _, h1, _ = torch_forward(blob, resi, 32, resi=True)
    out["resi_ctus"] = resi
    out["resi_seed_gain"] = np.array([3, 1.0])
    out["resi_vec"] = h1.astype(np.float32)
    # a blob checksum so a change of the generator is caught
    out["blob_a_crc"] = np.array([int(np.frombuffer(o.synth_blob(1, 1.0).tobytes(), dtype=np.uint32).sum(dtype=np.uint64))], dtype=np.uint64)
    path = os.path.join(HERE, "golden_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

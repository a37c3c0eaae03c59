"""Generates tests/golden/lstm_torch_golden.npz: the ETH-LSTM cell evaluated by a THIRD-PARTY implementation,
torch.nn.LSTMCell (float64), on the reference's TRAINED weights (HM-16.5_Test_LDP/bin/model_LDP_200000_qp{22,27,32,37}.dat).

Runs only in the build container (reads /root/reference); the .npz travels.  What it pins (task 4a): the
oracle's reading of tf.contrib.rnn.LSTMCell(n, forget_bias=1.0, cell_clip=5.0) at net_CNN_LSTM_one_step.py:205-206 --
gate blocks of the fused kernel in the order i, j, f, o; forget bias inside the sigmoid; clip of c before the output
gate -- against an independent cell whose gate order is i, f, g, o:

    torch weight_ih = K[:n, perm].T, weight_hh = K[n:, perm].T, perm = blocks (i, f, j, o) of TF's (i, j, f, o)
    torch bias_ih   = b[perm], bias_hh = +1.0 on the forget block (TF adds forget_bias inside sigmoid(f + 1))
    cell_clip       : LSTMCell has none, so the main set keeps |c| < 5 (asserted) and the clip set recovers the output
                      gate from the unclipped cell (o = h' / tanh(c')) and re-applies h = o * tanh(clip(c', -5, 5)).

If the oracle had the gate order wrong (e.g. PyTorch's own i, f, g, o read straight off the TF kernel) the states would
differ at the 1e-1 level on these weights; the test tolerance is 1e-5.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import ethcnn_lstm_np as ol  # tensor table only (names / shapes / offsets = the reference's .index)

REF = "/root/reference/HM-16.5_Test_LDP/bin"
N_MAIN, N_CLIP = 10, 6


def torch_cell(K, b, n):
    blk = lambda q: np.arange(q * n, (q + 1) * n)
    perm = np.concatenate([blk(0), blk(2), blk(1), blk(3)])  # TF (i, j, f, o) -> torch (i, f, g = j, o)
    cell = torch.nn.LSTMCell(n, n, bias=True).double()
    with torch.no_grad():
        cell.weight_ih.copy_(torch.from_numpy(K[:n, perm].T.astype(np.float64).copy()))
        cell.weight_hh.copy_(torch.from_numpy(K[n:, perm].T.astype(np.float64).copy()))
        cell.bias_ih.copy_(torch.from_numpy(b[perm].astype(np.float64)))
        fb = np.zeros(4 * n)
        fb[n:2 * n] = 1.0  # forget_bias=1.0
        cell.bias_hh.copy_(torch.from_numpy(fb))
    return cell


def run(blob, vec, state, clip):
    tv = ol.lstm_views(blob)
    out = np.zeros_like(state, dtype=np.float64)
    o1 = 0
    for tag, n in (("64", 64), ("32", 128), ("16", 256)):
        K = tv["RNN%s/multi_rnn_cell/cell_0/lstm_cell/kernel" % tag]
        b = tv["RNN%s/multi_rnn_cell/cell_0/lstm_cell/bias" % tag]
        cell = torch_cell(K, b, n)
        x = torch.from_numpy(vec[:, o1:o1 + n].astype(np.float64))
        c0 = torch.from_numpy(state[:, 0, o1:o1 + n].astype(np.float64))
        h0 = torch.from_numpy(state[:, 1, o1:o1 + n].astype(np.float64))
        with torch.no_grad():
            h1, c1 = cell(x, (h0, c0))
        h1, c1 = h1.numpy(), c1.numpy()
        if clip:
            o = h1 / np.tanh(c1)
            c1 = np.clip(c1, -5.0, 5.0)
            h1 = o * np.tanh(c1)
        else:
            assert np.abs(c1).max() < 5.0, "main set must not reach the cell clip"
        out[:, 0, o1:o1 + n], out[:, 1, o1:o1 + n] = c1, h1
        o1 += n
    return out


def main():
    rng = np.random.default_rng(20260928)
    vec_main = np.abs(rng.standard_normal((N_MAIN, 448))).astype(np.float32) * np.float32(0.7)     # FC1 outputs are mostly >= 0
    st_main = np.stack([rng.uniform(-2, 2, (N_MAIN, 448)), np.tanh(rng.standard_normal((N_MAIN, 448)))], 1).astype(np.float32)
    vec_clip = (rng.standard_normal((N_CLIP, 448)) * 2.0).astype(np.float32)
    st_clip = np.stack([rng.choice([-5.0, 5.0], (N_CLIP, 448)) * rng.uniform(0.9, 1.0, (N_CLIP, 448)),
                        np.tanh(rng.standard_normal((N_CLIP, 448)) * 2)], 1).astype(np.float32)
    out = {"vec_main": vec_main, "state_main": st_main, "vec_clip": vec_clip, "state_clip": st_clip}
    for qp in (22, 27, 32, 37):
        blob = np.fromfile(os.path.join(REF, "model_LDP_200000_qp%d.dat.data-00000-of-00001" % qp), dtype=np.float32)
        assert blob.size == ol.LSTM_BLOB_FLOATS
        out["out_main_qp%d" % qp] = run(blob, vec_main, st_main, False).astype(np.float32)
        oc = run(blob, vec_clip, st_clip, True)
        out["out_clip_qp%d" % qp] = oc.astype(np.float32)
        out["clipped_qp%d" % qp] = np.int64((np.abs(oc[:, 0]) >= 5.0).sum())
        print("qp%d: main |c| max %.3f, clip set: %d of %d cells at the clip" % (qp, np.abs(out["out_main_qp%d" % qp][:, 0]).max(),
                                                                                 out["clipped_qp%d" % qp], oc[:, 0].size))
    out["torch_version"] = np.array(torch.__version__)
    np.savez_compressed(os.path.join(HERE, "lstm_torch_golden.npz"), **out)


if __name__ == "__main__":
    main()

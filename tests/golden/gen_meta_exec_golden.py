#!/usr/bin/env python
"""Golden vectors from the reference's OWN serialized TensorFlow graphs, executed node by node by
tests/meta_graph.py (numpy, no TensorFlow) -- run HERE (needs /root/reference), commit the .npz.

  AI : HM-16.5_Test_AI/bin/model_2000000_qp30~35.dat.meta   x [n,64,64,1], qp -> y64|y32|y16 (ungated;
       the saved graph is the training script's: the threshold gates exist only in net_CNN.py:175,187)
       and h_conv_flat
  LDP: HM-16.5_Test_LDP/bin/model_LDP_2000000_qp22~37.dat.meta   residual CTUs -> the 448-vector
       [h_fc1_64 | h_fc1_32 | h_fc1_16] (resi_cnn, net_CNN_LSTM_one_step.py:151-199)

Weights: the seeded synthetic blob (oracle/ethcnn_np.py::synth_blob; the trained .data blobs are
absent from the reference) fed to the graph's VariableV2 nodes by name.  The file holds inputs,
seeds and outputs only (data, no reference source).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ethcnn_np as oracle  # noqa: E402  (only for the seeded weight generator + tensor table)
import meta_graph as mg  # noqa: E402


def ctus_ai(rng, n):
    c = rng.integers(0, 256, size=(n, 64, 64), dtype=np.uint8)
    yy, xx = np.mgrid[0:64, 0:64]
    k = n // 4
    c[:k] = ((yy * 2 + xx)[None] + rng.integers(0, 8, size=(k, 64, 64))).clip(0, 255).astype(np.uint8)
    c[k:2 * k] = rng.integers(0, 256, size=(k, 1, 1), dtype=np.uint8)
    c[2 * k], c[2 * k + 1] = 0, 255
    return c


def main():
    rng = np.random.default_rng(20260928)
    out = {}
    nodes = mg.load_nodes(mg.AI_META)
    for tag, seed, gain, qp, n in (("ai_a", 31, 1.0, 32, 40), ("ai_b", 32, 8.0, 22, 24)):
        blob = oracle.synth_blob(seed, gain)
        ctus = ctus_ai(rng, n)
        probs, feat, ops = mg.run_ai_graph(nodes, dict(oracle.tensor_views(blob)), ctus, qp)
        out[tag + "_seed_gain_qp"] = np.array([seed, gain, qp], dtype=np.float64)
        out[tag + "_ctus"] = ctus
        out[tag + "_probs"] = probs
        out[tag + "_feat8"] = feat[:8]
        print(tag, "ops executed:", sorted(ops))
    nodes = mg.load_nodes(mg.LDP_CNN_META)
    blob = oracle.synth_blob(33, 1.0)
    ctus = np.clip(128 + rng.laplace(0, 8, size=(20, 64, 64)), 0, 255).astype(np.uint8)
    ctus[0], ctus[1] = 128, 0
    vec, ops = mg.run_resi_graph(nodes, dict(oracle.tensor_views(blob)), ctus)
    out["ldp_seed_gain"] = np.array([33, 1.0], dtype=np.float64)
    out["ldp_ctus"] = ctus
    out["ldp_vec"] = vec
    print("ldp ops executed:", sorted(ops))
    path = os.path.join(HERE, "meta_exec_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

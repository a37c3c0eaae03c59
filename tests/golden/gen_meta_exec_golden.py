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
sys.path.insert(0, HERE)
import ctu_gen  # noqa: E402


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
    # large sets: inputs regenerated from (seed, n) by tests/golden/ctu_gen.py, only the outputs are stored
    nodes = mg.load_nodes(mg.AI_META)
    for tag, seed, gain, qp, gseed, n in (("ai_c", 34, 1.0, 32, 7001, 1024), ("ai_d", 35, 8.0, 27, 7002, 1024),
                                          ("ai_e", 36, 8.0, 37, 7003, 256)):
        blob = oracle.synth_blob(seed, gain)
        ctus = ctu_gen.make_ctus(gseed, n)
        probs = np.empty((n, 21), dtype=np.float32)
        feat8 = None
        for s0 in range(0, n, 128):
            p_, f_, _ = mg.run_ai_graph(nodes, dict(oracle.tensor_views(blob)), ctus[s0:s0 + 128], qp)
            probs[s0:s0 + 128] = p_
            if s0 == 0:
                feat8 = f_[:8]
        out[tag + "_seed_gain_qp"] = np.array([seed, gain, qp], dtype=np.float64)
        out[tag + "_gen"] = np.array([gseed, n, ctu_gen.crc(ctus)], dtype=np.int64)
        out[tag + "_probs"] = probs
        out[tag + "_feat8"] = feat8
        print(tag, n, "CTUs, p range", probs.min(), probs.max())
    nodes = mg.load_nodes(mg.LDP_CNN_META)
    blob = oracle.synth_blob(37, 1.0)
    ctus = ctu_gen.make_ctus(7004, 256, residual=True)
    vec, _ = mg.run_resi_graph(nodes, dict(oracle.tensor_views(blob)), ctus)
    out["ldp_b_seed_gain"] = np.array([37, 1.0], dtype=np.float64)
    out["ldp_b_gen"] = np.array([7004, 256, ctu_gen.crc(ctus)], dtype=np.int64)
    out["ldp_b_vec"] = vec
    path = os.path.join(HERE, "meta_exec_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

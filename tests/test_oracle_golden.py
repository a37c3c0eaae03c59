"""CPU: pins the oracle.  (1) against the committed golden vectors of the independent
PyTorch restatement (tests/golden/gen_golden.py), (2) against the independent numpy float64
restatement computed live, (3) its constants / wiring against the reference's own .meta
graph (tests/golden/meta_constants.json), (4) canonical vs literal summation order.
Tolerance: the north star's 1e-4 on probabilities (measured ~1e-6); decisions identical."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "golden_v1.npz"))
META = json.load(open(os.path.join(HERE, "golden", "meta_constants.json")))["nodes"]
TOL = 1e-4


def _run(oracle, blob, ctus, qp, mode, resi=0):
    F = oracle.features(blob, ctus, mode, resi)
    H1 = oracle.fc1(blob, F, mode)
    P, Z = oracle.heads(blob, H1, qp, mode)
    return F, H1, P, Z


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("mode", [0, 1])
def test_oracle_vs_torch_golden(oracle, tag, mode):
    seed, gain, qp = G[tag + "_seed_gain_qp"]
    blob = oracle.synth_blob(int(seed), float(gain))
    if tag == "a":  # the fixture was generated with exactly this weight blob
        assert int(np.frombuffer(blob.tobytes(), dtype=np.uint32).sum(dtype=np.uint64)) == int(G["blob_a_crc"][0])
    F, H1, P, _ = _run(oracle, blob, G["ctus"], int(qp), mode)
    assert np.abs(P - G[tag + "_probs"]).max() <= TOL
    assert np.abs(F[:4] - G[tag + "_feat4"]).max() <= 2e-5
    assert np.abs(H1[:4] - G[tag + "_h1_4"]).max() <= 2e-5
    far = np.abs(G[tag + "_probs"] - 0.5) > 1e-4  # decisions, away from a knife edge
    assert np.array_equal((P > 0.5)[far], (G[tag + "_probs"] > 0.5)[far])


def test_oracle_resi_vs_torch_golden(oracle):
    seed, gain = G["resi_seed_gain"]
    blob = oracle.synth_blob(int(seed), float(gain))
    for mode in (0, 1):
        V = oracle.fc1(blob, oracle.features(blob, G["resi_ctus"], mode, 1), mode)
        assert np.abs(V - G["resi_vec"]).max() <= 2e-5


@pytest.mark.parametrize("gain,qp", [(1.0, 32), (8.0, 37)])
def test_oracle_vs_float64_restatement(oracle, gain, qp):
    rng = np.random.default_rng(17)
    ctus = rng.integers(0, 256, size=(40, 64, 64), dtype=np.uint8)
    ctus[:10] //= 8
    blob = oracle.synth_blob(9, gain)
    ref = oracle.forward64(blob, ctus, qp)
    for mode in (0, 1):
        F, H1, P, Z = _run(oracle, blob, ctus, qp, mode)
        assert np.abs(F - ref["F"]).max() <= 5e-6
        assert np.abs(H1 - ref["H1"]).max() <= 5e-6
        assert np.abs(P - ref["probs"]).max() <= TOL
    F0, _, P0, _ = _run(oracle, blob, ctus, qp, 0)
    F1, _, P1, _ = _run(oracle, blob, ctus, qp, 1)
    assert np.abs(F0 - F1).max() <= 5e-6 and np.abs(P0 - P1).max() <= 2e-5  # order = rounding-level only


def test_unit_decomposition_and_feature_map(oracle):
    """Moving one 16x16 block's pixels changes exactly that S unit's 128 features plus the
    enclosing M unit's and the L unit's."""
    rng = np.random.default_rng(3)
    blob = oracle.synth_blob(4, 1.0)
    a = rng.integers(0, 256, size=(1, 64, 64), dtype=np.uint8)
    b = a.copy()
    b[0, 16:32, 48:64] = rng.integers(0, 256, size=(16, 16))  # S unit (by=1, bx=3) -> M unit (0,1)
    Fa, Fb = oracle.features(blob, a, 0)[0], oracle.features(blob, b, 0)[0]
    changed = np.nonzero(Fa != Fb)[0]
    allowed = set()
    allowed |= set(range((1 * 4 + 3) * 32, (1 * 4 + 3) * 32 + 32))                      # conv3_S
    allowed |= set(range(512 + (0 * 2 + 1) * 32, 512 + (0 * 2 + 1) * 32 + 32))          # conv3_M
    allowed |= set(range(640, 672))                                                      # conv3_L
    for y in (2, 3):
        for x in (6, 7):
            allowed |= set(range(672 + (y * 8 + x) * 24, 672 + (y * 8 + x) * 24 + 24))   # conv2_S
    for y in (0, 1):
        for x in (2, 3):
            allowed |= set(range(2208 + (y * 4 + x) * 24, 2208 + (y * 4 + x) * 24 + 24))  # conv2_M
    allowed |= set(range(2592, 2688))                                                    # conv2_L
    assert set(changed.tolist()) <= allowed and len(changed) > 200


def test_gates_semantics(oracle):
    P = np.full((2050, 21), 0.4, dtype=np.float32)
    P[5, 0] = 0.9          # chunk 0: L1 open; no y32 above thr2 -> y16 zeroed
    P[1500, 0] = 0.9       # chunk 1: L1 open, a y32 above thr2 -> all kept
    P[1501, 2] = 0.8
    out = oracle.gates(P, 0.5, 0.5)
    assert out[:1024, 1:5].all() and not out[:1024, 5:].any()
    assert out[1024:2048].all()
    assert not out[2048:, 1:].any() and (out[2048:, 0] == np.float32(0.4)).all()  # chunk 2 (2 CTUs): closed
    out = oracle.gates(P, 0.95, -1.0)  # everything closed at L1, but zeros > -1 keeps y16
    assert not out[:, 1:5].any() and out[:, 5:].all()
    out = oracle.gates(np.full((4, 21), 0.5, np.float32), 0.5, 0.5)  # strict '>' (net_CNN.py:175)
    assert not out[:, 1:].any()


def test_tiling_matches_reference_padding(oracle):
    """video_to_cu_depth.py:46-59,94-104: zero pad bottom/right, raster order."""
    rng = np.random.default_rng(1)
    for (w, h) in ((200, 136), (64, 64), (65, 1), (1920, 1080)):
        luma = rng.integers(1, 256, size=(h, w), dtype=np.uint8)
        vh, vw = -(-h // 64) * 64, -(-w // 64) * 64
        padded = np.zeros((vh, vw), np.uint8)
        padded[:h, :w] = luma
        want = padded.reshape(vh // 64, 64, vw // 64, 64).transpose(0, 2, 1, 3).reshape(-1, 64, 64)
        assert np.array_equal(oracle.tile_frame(luma, w, h), want)


def test_predict_frames_equals_per_ctu_path(oracle):
    rng = np.random.default_rng(8)
    blob = oracle.synth_blob(2, 8.0)
    w, h = 64 * 36, 64 * 30  # 1080 CTUs: sub-batches 1024 + 56
    luma = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    got = oracle.predict_frames(blob, luma, w, h, 1, 27, 0.5, 0.5)
    ctus = oracle.tile_frame(luma, w, h)
    _, _, P, _ = _run(oracle, blob, ctus, 27, 0)
    assert np.array_equal(got, oracle.gates(P, 0.5, 0.5, chunk=1024))


# ---- the reference's own graph (.meta) ---------------------------------------------------
def test_meta_constants():
    f32 = lambda x: "0x%08x" % np.float32(x).view(np.uint32)
    assert META["scalar"]["bits"] == [f32(1.0 / 255.0)]            # net_CNN.py:105
    assert META["scalar_1"]["bits"] == [f32(1.0 / 51.0)]           # :106
    alphas = [v for k, v in META.items() if k.endswith("/alpha") and k.startswith("LeakyRelu")]
    assert len(alphas) >= 15 and all(a["bits"] == [f32(0.2)] for a in alphas)
    mean_kernels = [v for v in META.values() if v["op"] == "Const" and v.get("shape") == [16, 16, 1, 1]]
    assert len(mean_kernels) == 3 and all(m["bits"] == [f32(1.0 / 256.0)] for m in mean_kernels)


def test_meta_conv_pool_wiring():
    convs = {k: v for k, v in META.items() if v["op"] == "Conv2D"}
    mean = [v for v in convs.values() if v["strides"] == [1, 16, 16, 1]]
    assert len(mean) == 3 and all(v["padding"] == "VALID" for v in mean)
    feat = sorted((v["strides"][1] for v in convs.values() if v["strides"] != [1, 16, 16, 1]))
    assert feat == [2] * 6 + [4] * 3 and all(v["padding"] == "VALID" for v in convs.values())
    sizes = sorted(tuple(v["values"]) for k, v in META.items() if k.startswith("ResizeNearestNeighbor") and k.endswith("/size"))
    assert sizes == [(16, 16), (32, 32), (64, 64)]
    pools_x = [v for v in META.values() if v["op"] == "AvgPool" and v["inputs"] == ["Reshape"]]  # fed by x_image
    assert sorted(v["ksize"][1] for v in pools_x) == [2, 4] and all(v["padding"] == "SAME" for v in pools_x)


def test_meta_concat_order_and_qp_last():
    cat = META["concat"]                                            # h_conv_flat, net_CNN.py:150
    widths = [META[i + "/shape"]["values"][1] for i in cat["inputs"][:-1]]
    assert widths == [512, 128, 32, 1536, 384, 96]                  # c3S c3M c3L c2S c2M c2L
    fc_cats = [v for k, v in META.items() if v["op"] == "ConcatV2" and k != "concat" and len(v["inputs"]) == 3]
    assert len(fc_cats) == 6 and all(v["inputs"][1] == "mul_1" for v in fc_cats)   # [h, qp]: qp LAST
    assert META["mul_1"]["inputs"] == ["scalar_1", "Placeholder_2"]


def test_blocked_fc1_is_bit_identical_to_the_per_ctu_chain(oracle):
    """oracle_predict_frames / oracle_resi_vectors run FC1 for 8 CTUs at a time (W1 tiles reused, accumulators in
    registers); the staged per-CTU functions are the plain restatement.  Same chains, same bits, both modes, ragged
    block tails (n % 8 != 0)."""
    import bench
    w, h = 64 * 7, 64 * 3 - 9  # 21 CTUs: two full blocks + a tail of 5
    luma = bench.synth_luma(w, h, 2, seed=12)
    blob = oracle.synth_blob(7, 8.0)
    for mode in (0, 1):
        whole = oracle.predict_frames(blob, luma, w, h, 2, 27, 0.5, 0.5, mode=mode)
        for f in range(2):
            ctus = oracle.tile_frame(luma[f], w, h)
            P, _ = oracle.heads(blob, oracle.fc1(blob, oracle.features(blob, ctus, mode=mode), mode), 27, mode)
            want = oracle.gates(P, 0.5, 0.5)
            assert np.array_equal(whole[f * 21:(f + 1) * 21].view(np.uint32), want.view(np.uint32))
        ctus = oracle.tile_frame(luma[0], w, h)
        V = oracle.resi_vectors(blob, luma[0], w, h, mode=mode)
        assert np.array_equal(V.view(np.uint32), oracle.fc1(blob, oracle.features(blob, ctus, mode=mode, resi=1), mode).view(np.uint32))

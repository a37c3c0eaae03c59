"""CPU: the oracle against what THE REFERENCE'S OWN PYTHON FILES wrote when they were executed in the build
container over tests/tf_shim.py (tests/golden/gen_ref_exec_golden.py; tests/ref_exec.py runs them):
video_to_cu_depth.py as __main__ with its real argv -> cu_depth.dat, and resi_to_cu_depth_LDP.py's daemon
loop driven over its file protocol -> cu_depth.dat + state.dat per frame.

Covered with the reference's own lines (and by no .meta graph): get_Y_for_one_frame's zero pad, the tiling loop and
1024-CTU sub-batching (video_to_cu_depth.py:46-118), the QP-band restore (:126-133), both batch gates
(net_CNN.py:175,187), the LDP net() wiring incl. efs / one-hot / state slicing (net_CNN_LSTM_one_step.py:201-323),
get_images_from_one_file / get_state_in_from_one_file / predict_cu_depth (resi_to_cu_depth_LDP.py:72-129).
NOT covered: TensorFlow's own op kernels.  Two stand-ins for them, one fixture each (tests/tf_shim.py, TF_SHIM_KERNELS):
  numpy  the repo's restatements, float64 accumulation rounded once (ref_exec_golden.npz)        -> the oracle within 1e-5
  torch  PYTORCH'S OWN float32 CPU kernels for Conv2D / AvgPool / MatMul / Sigmoid / Tanh / ResizeNearestNeighbor
         (ref_exec_golden_torch.npz): the reference's program over a third party's arithmetic, nothing of it written
         here; its fp32 accumulation order differs from the canonical one as TensorFlow's would -> within 3e-5
The north star's bar is 1e-4; gate patterns are identical in every case.
"""
import os
import sys

import numpy as np
import pytest

from conftest import have_reference

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import gen_ref_exec_golden as gen  # noqa: E402  (input / weight regeneration by seed; no reference access at import)

GOLDEN = os.path.join(HERE, "golden", "ref_exec_golden.npz")
GOLDEN_TORCH = os.path.join(HERE, "golden", "ref_exec_golden_torch.npz")
TOL = 1e-5          # canonical order (mode 0: what the kernels compute); measured <= 2.4e-6
TOL_LITERAL = 5e-5  # literal plain-fp32 chains (mode 1) under the x8 head gain (logits to +-20); measured <= 2e-5
TOL_TORCH = 3e-5    # either mode against torch's own fp32 kernels (order noise on both sides); measured <= 1.5e-5
KERNEL_SETS = ["numpy", "torch"]


def tol_for(g, mode=0):
    if mode == 1:
        return TOL_LITERAL
    return TOL_TORCH if str(g["kernels"]) == "torch" else TOL


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLDEN)


@pytest.fixture(scope="module", params=KERNEL_SETS)
def gold_k(request):
    """the fixture of either stand-in kernel set"""
    return np.load(GOLDEN if request.param == "numpy" else GOLDEN_TORCH)


def thr13(gold, tag):
    """tokens [1] and [3] of the Thr_info.txt line the reference run read (net_CNN.py:38-47)"""
    tok = str(gold[tag + "_thr"]).split(" ")
    return float(tok[1]), float(tok[3])


def ai_case(gold, tag):
    w, h, nf, qp = (int(v) for v in gold[tag + "_whfq"])
    if tag == "ai_small":
        frames = gen.ai_small_frames()
    elif tag.startswith("ai_qp"):
        frames = [gen.ai_qp_frame(qp)]
    else:
        frames = [gen.big_frame()]
    assert len(frames) == nf and frames[0].shape == (h, w)
    return w, h, nf, qp, np.stack(frames), gen.blobs()[gen.band_of(qp)]


AI_TAGS = ["ai_small"] + ["ai_qp%d" % q for q in gen.QPS] + ["ai_big_open", "ai_big_l1", "ai_big_l2"]


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("tag", AI_TAGS)
def test_oracle_matches_the_reference_scripts_output_ai(oracle, gold_k, tag, mode):
    gold = gold_k
    w, h, nf, qp, luma, blob = ai_case(gold, tag)
    t1, t2 = thr13(gold, tag)
    want = gold[tag + "_probs"]
    got = oracle.predict_frames(blob, luma, w, h, nf, qp, t1, t2, mode=mode)
    assert got.shape == want.shape
    assert np.array_equal(got == 0, want == 0), "gate pattern differs from the reference run"
    assert np.abs(got - want).max() <= tol_for(gold, mode)


def test_the_two_stand_in_kernel_sets_agree(gold):
    """numpy restatements (float64 accumulation) against torch's fp32 kernels under the reference's program: same cases,
    same Thr_info.txt texts, same gate patterns, values within fp32 order noise"""
    t = np.load(GOLDEN_TORCH)
    assert str(t["kernels"]) == "torch" and sorted(k for k in gold.files if k != "kernels") == sorted(k for k in t.files if k != "kernels")
    assert set(gold["ops_executed"].tolist()) == set(t["ops_executed"].tolist())
    for k in gold.files:
        if k.endswith("_thr"):
            assert str(gold[k]) == str(t[k]), k
        if k.endswith("_probs") or k.endswith("_state"):
            assert np.array_equal(gold[k] == 0, t[k] == 0), k
            assert np.abs(gold[k].astype(np.float64) - t[k]).max() <= 2.5e-5, k


def test_fixture_covers_what_it_claims(gold_k):
    gold = gold_k
    # sub-batching: 1200 CTUs = 1024 + 176, the gate states differ between the two sub-batches of one frame
    p1, p2 = gold["ai_big_l1_probs"], gold["ai_big_l2_probs"]
    assert p1.shape == (1200, 21)
    assert (p1[:1024, 1:] != 0).all() and (p1[1024:, 1:] == 0).all() and (p1[:, 0] != 0).all()
    assert (p2[:, :5] != 0).all() and (p2[:1024, 5:] != 0).all() and (p2[1024:, 5:] == 0).all()
    assert (gold["ai_big_open_probs"] != 0).all()
    # the model switch: same frame generator, another band's blob -> the wrong restore is far outside the tolerance
    import ethcnn_np as oracle
    for qp in (24, 25, 29, 30, 34, 35):
        w, h, nf, q, luma, blob = ai_case(gold, "ai_qp%d" % qp)
        other = gen.blobs()[gen.band_of(qp + (1 if qp % 5 == 4 else -1))]
        wrong = oracle.predict_frames(other, luma, w, h, nf, qp, 0.5, 0.5)
        assert np.abs(wrong - gold["ai_qp%d_probs" % qp]).max() > 0.05
    # LDP: every reachable gate state occurs over the two threshold sets; i_frame % 4 wraps
    states = set()
    for tag in ("ldp_a", "ldp_b"):
        for P in gold[tag + "_probs"]:
            states.add((bool((P[:, 1:5] != 0).any()), bool((P[:, 5:] != 0).any())))
    assert states == {(False, False), (True, False), (True, True)}
    assert [int(i) % 4 for i in gold["ldp_cfg"][5:]] == [1, 2, 3, 0, 1]
    assert {"cond", "count_nonzero", "OneHot", "ClipByValue", "Tanh", "Conv2D", "MatMul"} <= set(gold["ops_executed"].tolist())


def ldp_inputs(gold):
    w, h, qp, cnn_seed, fseed = (int(v) for v in gold["ldp_cfg"][:5])
    assert (w, h, qp, cnn_seed, fseed) == (gen.LDP["w"], gen.LDP["h"], gen.LDP["qp"], gen.LDP["cnn_seed"], gen.LDP["frame_seed"])
    import ethcnn_np as oracle
    lstm = np.fromfile(os.path.join(HERE, "golden", "model_LDP_200000_qp32.dat.data-00000-of-00001"), dtype="<f4")
    return w, h, qp, oracle.synth_blob(cnn_seed, gen.LDP["cnn_gain"]), lstm, gen.ldp_frames(), [int(i) for i in gold["ldp_cfg"][5:]]


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("tag", ["ldp_a", "ldp_b"])
def test_oracle_matches_the_reference_daemon_ldp(oracle, gold_k, tag, mode):
    """resi_cnn -> one ETH-LSTM step -> heads -> gates, the state fed back frame to frame (the oracle's own state, not
    the fixture's: errors would accumulate over the recurrence if there were any)"""
    import ethcnn_lstm_np as ol
    gold = gold_k
    w, h, qp, cnn, lstm, frames, i_frames = ldp_inputs(gold)
    t1, t2 = thr13(gold, tag)
    state = None
    for k, (luma, i_frame) in enumerate(zip(frames, i_frames)):
        V = oracle.resi_vectors(cnn, luma, w, h, mode=mode)
        P, S = ol.lstm_step(lstm, V, None if i_frame <= 1 else state, qp, i_frame, t1, t2, mode=mode)
        want = gold[tag + "_probs"][k]
        assert np.array_equal(P == 0, want == 0), (tag, i_frame)
        assert np.abs(P - want).max() <= tol_for(gold), (tag, i_frame)
        if tag == "ldp_a":
            assert np.abs(S.reshape(-1) - gold["ldp_a_state"][k].reshape(-1)).max() <= max(tol_for(gold), TOL if mode == 0 else 2e-5), i_frame
        state = S


# ---- live: the reference's files executed again, here (build container only) ---------------------------------
@pytest.mark.skipif(not have_reference(), reason="needs /root/reference (build container only)")
def test_live_reference_ai_script_reproduces_the_fixture(gold, tmp_path):
    import ctu_gen
    import ref_exec
    P, rep, text = ref_exec.run_ai(str(tmp_path / "ai"), ctu_gen.yuv420_bytes(gen.ai_small_frames()), 200, 136, 32,
                                   str(gold["ai_small_thr"]), gen.blobs())
    assert np.array_equal(P, gold["ai_small_probs"])
    assert rep["restored"] == [["model_2000000_qp30~35.dat", 36]] and "Predicting Time:" in text
    # the dropout branches and the label plumbing stay dead at isdrop = 0
    assert not {"Dropout", "Relu"} & set(rep["ops"])
    # two sub-batches, L1 closed in the second one only: re-run the real thing on the big frame
    P, rep, _ = ref_exec.run_ai(str(tmp_path / "big"), ctu_gen.yuv420_bytes([gen.big_frame()]), 2560, 1920, 32,
                                str(gold["ai_big_l1_thr"]), gen.blobs())
    assert np.array_equal(P, gold["ai_big_l1_probs"])


@pytest.mark.skipif(not have_reference(), reason="needs /root/reference (build container only)")
def test_live_reference_scripts_over_torch_kernels_reproduce_the_torch_fixture(tmp_path):
    """both scripts again with PyTorch's kernels inside the tf.* calls (bits may move with the host's ISA / thread
    count, so within 1e-5 of the committed run rather than equal)"""
    import ctu_gen
    import ref_exec
    t = np.load(GOLDEN_TORCH)
    P, rep, _ = ref_exec.run_ai(str(tmp_path / "ai"), ctu_gen.yuv420_bytes(gen.ai_small_frames()), 200, 136, 32,
                                str(t["ai_small_thr"]), gen.blobs(), kernels="torch")
    assert rep["kernels"] == "torch" and np.abs(P - t["ai_small_probs"]).max() <= 1e-5
    w, h, qp, cnn, lstm32, frames, i_frames = ldp_inputs(t)
    dm = ref_exec.LdpDaemon(str(tmp_path / "ldp"), str(t["ldp_b_thr"]), cnn,
                            {"model_LDP_200000_qp32.dat": os.path.join(ref_exec.REF_LDP_BIN, "model_LDP_200000_qp32.dat")}, kernels="torch")
    try:
        for k in range(3):
            P, S = dm.frame(frames[k], i_frames[k], qp)
            assert np.array_equal(P == 0, t["ldp_b_probs"][k] == 0) and np.abs(P - t["ldp_b_probs"][k]).max() <= 1e-5
            assert np.abs(S - t["ldp_a_state"][k]).max() <= 2e-5
    finally:
        rep = dm.close()
    assert rep["kernels"] == "torch"


@pytest.mark.skipif(not have_reference(), reason="needs /root/reference (build container only)")
def test_live_reference_ldp_daemon_reproduces_the_fixture_and_switches_models(gold, oracle, tmp_path):
    """the daemon again, three frames; then a QP change mid-stream: it restores another band's LSTM bundle
    (resi_to_cu_depth_LDP.py:166-179) -- checked against the oracle fed that bundle"""
    import ethcnn_lstm_np as ol
    import ref_exec
    w, h, qp, cnn, lstm32, frames, i_frames = ldp_inputs(gold)
    names = {"model_LDP_200000_qp%d.dat" % q: os.path.join(ref_exec.REF_LDP_BIN, "model_LDP_200000_qp%d.dat" % q) for q in (32, 37)}
    dm = ref_exec.LdpDaemon(str(tmp_path / "ldp"), str(gold["ldp_a_thr"]), cnn, names)
    try:
        for k in range(3):
            P, S = dm.frame(frames[k], i_frames[k], qp)
            assert np.array_equal(P, gold["ldp_a_probs"][k]) and np.array_equal(S, gold["ldp_a_state"][k])
        P, S37 = dm.frame(frames[3], i_frames[3], 37)
    finally:
        rep = dm.close()
    assert [r[0] for r in rep["restored"]] == [ref_exec.LDP_CNN_NAME, "model_LDP_200000_qp32.dat", "model_LDP_200000_qp37.dat"]
    lstm37 = np.fromfile(names["model_LDP_200000_qp37.dat"] + ".data-00000-of-00001", dtype="<f4")
    t1, t2 = thr13(gold, "ldp_a")
    V = oracle.resi_vectors(cnn, frames[3], w, h)
    OP, OS = ol.lstm_step(lstm37, V, gold["ldp_a_state"][2].reshape(-1, 2, 448), 37, i_frames[3], t1, t2)
    assert np.array_equal(OP == 0, P == 0) and np.abs(OP - P).max() <= TOL
    assert np.abs(OS.reshape(-1) - S37.reshape(-1)).max() <= TOL

"""CPU: the ETH-LSTM one-step oracle (oracle_lstm_step) against an independent numpy float64
restatement of net_CNN_LSTM_one_step.py:201-323, on synthetic weights and on the reference's own
trained weights (tests/golden/model_LDP_200000_qp32.dat.* = the data files shipped in
/root/reference/HM-16.5_Test_LDP/bin), plus the host-only pieces of the LDP row (bundle reader,
model-name bands, synthetic generator).  The reference's TF-1.x graph cannot run here (no
TensorFlow), so this row is "parity unpinned" against TF outputs: see DESIGN.md."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REAL = os.path.join(GOLDEN, "model_LDP_200000_qp32.dat")


def _inputs(rng, n, scale=1.0):
    vec = (np.abs(rng.standard_normal((n, 448))) * scale).astype(np.float32)  # FC1 outputs are leaky-ReLU'd
    vec[:, ::7] *= -0.2
    state = np.stack([rng.uniform(-5, 5, (n, 448)), rng.uniform(-1, 1, (n, 448))], 1).astype(np.float32)
    return vec, state


@pytest.fixture(scope="module")
def lstm(oracle):
    import ethcnn_lstm_np
    return ethcnn_lstm_np


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("with_state", [False, True])
def test_oracle_vs_float64_synthetic(lstm, mode, with_state):
    rng = np.random.default_rng(3)
    blob = lstm.synth_lstm_blob(7, 2.0)
    vec, state = _inputs(rng, 45)
    P, S = lstm.lstm_step(blob, vec, state if with_state else None, 27, 6, thr1=-1.0, thr2=-1.0, mode=mode)
    rP, rS = lstm.lstm_forward64(blob, vec, state if with_state else None, 27, 6)
    assert np.abs(P - rP).max() <= 2e-6
    assert np.abs(S - rS).max() <= 5e-6
    assert np.abs(S[:, 0]).max() <= 5.0  # cell_clip


def test_oracle_vs_float64_trained_weights(pkg, lstm):
    blob = pkg.ethcnn.read_ckpt_lstm_blob(REAL)  # crc32c-checked by the product's bundle reader
    raw = np.fromfile(REAL + ".data-00000-of-00001", dtype=np.float32)
    assert np.array_equal(blob.view(np.uint32), raw.view(np.uint32))  # 18 tensors tile the payload exactly
    rng = np.random.default_rng(5)
    vec, state = _inputs(rng, 64, 0.5)
    for i_frame, sin in ((1, None), (2, state), (7, state)):
        P, S = lstm.lstm_step(blob, vec, sin, 32, i_frame, thr1=-1.0, thr2=-1.0, mode=0)
        rP, rS = lstm.lstm_forward64(blob, vec, sin, 32, i_frame)
        assert np.abs(P - rP).max() <= 2e-6 and np.abs(S - rS).max() <= 5e-6
        assert 0.0 < P.min() and P.max() < 1.0
    # the recurrence: feeding state_out back changes the prediction (state is really used)
    P1, S1 = lstm.lstm_step(blob, vec, None, 32, 1, -1.0, -1.0)
    P2, _ = lstm.lstm_step(blob, vec, S1, 32, 2, -1.0, -1.0)
    P2z, _ = lstm.lstm_step(blob, vec, None, 32, 2, -1.0, -1.0)
    assert not np.array_equal(P2, P2z)


def test_oracle_gates_per_mini_batch(lstm):
    """y32 zeroed unless any y64 > thr1 in the 1024-CTU mini-batch; y16 from the GATED y32
    (net_CNN_LSTM_one_step.py gates, resi_to_cu_depth_LDP.py:118 mini_batch_size)."""
    rng = np.random.default_rng(9)
    blob = lstm.synth_lstm_blob(4, 6.0)
    vec, _ = _inputs(rng, 1024 + 40)
    raw, _ = lstm.lstm_step(blob, vec, None, 37, 3, -1.0, -1.0)
    lo, hi = float(raw[:, 0].min()), float(raw[:, 0].max())
    thr1 = float(raw[1024:, 0].max())  # second mini-batch closed, first open (if it holds a larger value)
    if raw[:1024, 0].max() > thr1:
        P, _ = lstm.lstm_step(blob, vec, None, 37, 3, thr1, 0.5)
        assert np.array_equal(P[:1024, 1:5], raw[:1024, 1:5])
        assert not P[1024:, 1:].any()
    P, _ = lstm.lstm_step(blob, vec, None, 37, 3, hi, 0.5)
    assert np.array_equal(P[:, 0], raw[:, 0]) and not P[:, 1:].any()
    P, _ = lstm.lstm_step(blob, vec, None, 37, 3, lo - 1.0, 2.0)
    assert np.array_equal(P[:, :5], raw[:, :5]) and not P[:, 5:].any()


def test_lstm_index_is_the_table(pkg, lstm):
    ents = pkg.ethcnn.read_ckpt_index(REAL + ".index")
    assert [(e[0], e[2], e[4]) for e in ents] == [(n, tuple(s), o) for n, s, o in lstm.LSTM_TENSORS]
    assert sum(e[5] for e in ents) == lstm.LSTM_BLOB_BYTES == pkg.ethcnn.LSTM_BLOB_FLOATS * 4


def test_lstm_model_bands(pkg):
    f = pkg.ethcnn.lstm_model_name_for_qp  # resi_to_cu_depth_LDP.py:170-177
    assert [f(q) for q in (22, 24, 25, 29, 30, 34, 35, 51)] == [
        "model_LDP_200000_qp22.dat", "model_LDP_200000_qp22.dat", "model_LDP_200000_qp27.dat",
        "model_LDP_200000_qp27.dat", "model_LDP_200000_qp32.dat", "model_LDP_200000_qp32.dat",
        "model_LDP_200000_qp37.dat", "model_LDP_200000_qp37.dat"]


def test_lstm_bundle_errors(pkg, tmp_path):
    e = pkg.ethcnn
    with pytest.raises(e.EthCnnError):
        e.read_ckpt_lstm_blob(str(tmp_path / "nope.dat"))
    # a CNN index is not an LSTM bundle
    import shutil
    shutil.copy(os.path.join(GOLDEN, "model_2000000_qp30_35.dat.index"), tmp_path / "m.dat.index")
    (tmp_path / "m.dat.data-00000-of-00001").write_bytes(b"\0" * 64)
    with pytest.raises(e.EthCnnError) as ei:
        e.read_ckpt_lstm_blob(str(tmp_path / "m.dat"))
    assert "RNN16" in str(ei.value)
    # corrupt payload -> crc mismatch
    shutil.copy(REAL + ".index", tmp_path / "c.dat.index")
    raw = bytearray(open(REAL + ".data-00000-of-00001", "rb").read())
    raw[1000] ^= 0x40
    (tmp_path / "c.dat.data-00000-of-00001").write_bytes(bytes(raw))
    with pytest.raises(e.EthCnnError) as ei:
        e.read_ckpt_lstm_blob(str(tmp_path / "c.dat"))
    assert "crc32c" in str(ei.value)


def test_trained_weights_fit_the_i_j_f_o_gate_order(lstm):
    """Circumstantial check of TF's LSTMCell gate layout (i, j, f, o = input, candidate, forget, output
    quarters of the fused kernel/bias) on the reference's trained weights: in all three cells the
    candidate quarter (the only tanh branch) has clearly the largest bias spread and kernel column norm,
    and the forget quarter (which gets +1.0 at run time) the most negative mean bias.  With PyTorch's
    (i, f, g, o) layout the odd quarter out would be the third, not the second."""
    blob = np.fromfile(REAL + ".data-00000-of-00001", dtype=np.float32)
    tv = lstm.lstm_views(blob)
    for tag, n in (("64", 64), ("32", 128), ("16", 256)):
        b = tv["RNN%s/multi_rnn_cell/cell_0/lstm_cell/bias" % tag]
        K = tv["RNN%s/multi_rnn_cell/cell_0/lstm_cell/kernel" % tag]
        std = [float(b[i * n:(i + 1) * n].std()) for i in range(4)]
        norm = [float(np.linalg.norm(K[:, i * n:(i + 1) * n], axis=0).mean()) for i in range(4)]
        mean = [float(b[i * n:(i + 1) * n].mean()) for i in range(4)]
        assert int(np.argmax(std)) == 1 and int(np.argmax(norm)) == 1, (tag, std, norm)
        assert int(np.argmin(mean)) == 2, (tag, mean)

"""The oracle's ETH-LSTM cell against a THIRD-PARTY cell: torch.nn.LSTMCell (float64) on the reference's TRAINED weights
(tests/golden/lstm_torch_golden.npz, made by tests/golden/gen_lstm_torch_golden.py in the build container).

Pins what no reference-derived fixture pinned before (VERDICT r02 "missing" #2): TF LSTMCell's gate order i, j, f, o in the
fused kernel, forget_bias = 1 inside the sigmoid, cell_clip = 5 applied to c before the output gate
(net_CNN_LSTM_one_step.py:205-206).  Tolerance 1e-5 on (c, h); a wrong gate order is a 1e-1 error on these weights.
CPU: oracle_lstm_step (canonical + literal order) and the numpy float64 restatement; qp 32 always (its blob is a test
fixture), qp 22 / 27 / 37 where /root/reference is present.  GPU (-m gpu): the HIP ETH-LSTM step against the same file."""
import os

import numpy as np
import pytest

from conftest import REFERENCE

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-5


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "lstm_torch_golden.npz"))


@pytest.fixture(scope="module")
def lstm(oracle):
    import ethcnn_lstm_np
    return ethcnn_lstm_np


def _blob(qp):
    if qp == 32:
        return np.fromfile(os.path.join(GOLDEN, "model_LDP_200000_qp32.dat.data-00000-of-00001"), dtype=np.float32)
    p = os.path.join(REFERENCE, "HM-16.5_Test_LDP", "bin", "model_LDP_200000_qp%d.dat.data-00000-of-00001" % qp)
    if not os.path.exists(p):
        pytest.skip("trained qp%d LSTM blob lives only in /root/reference" % qp)
    return np.fromfile(p, dtype=np.float32)


@pytest.mark.parametrize("qp", [22, 27, 32, 37])
@pytest.mark.parametrize("which", ["main", "clip"])
def test_oracle_cell_matches_torch_lstmcell(lstm, gold, qp, which):
    blob = _blob(qp)
    vec, st, want = gold["vec_" + which], gold["state_" + which], gold["out_%s_qp%d" % (which, qp)]
    for mode in (0, 1):
        _, got = lstm.lstm_step(blob, vec, st, qp, 2, 0.5, 0.5, mode=mode)
        assert np.abs(got - want).max() <= TOL, (qp, which, mode, float(np.abs(got - want).max()))
    _, f64 = lstm.lstm_forward64(blob, vec, st, qp, 2)
    assert np.abs(f64 - want).max() <= 2e-6  # float64 vs float64 torch, outputs stored as float32
    if which == "clip":
        assert int(gold["clipped_qp%d" % qp]) > 50  # the clip really acts in this set
        assert np.abs(want[:, 0]).max() == 5.0


def test_a_wrong_gate_order_would_be_caught(lstm, gold):
    """sensitivity of the fixture: PyTorch's own block order (i, f, g, o) read straight off the TF kernel is far outside TOL"""
    blob = _blob(32).copy()
    tv = lstm.lstm_views(blob)
    for tag, n in (("64", 64), ("32", 128), ("16", 256)):
        for name in ("kernel", "bias"):
            t = tv["RNN%s/multi_rnn_cell/cell_0/lstm_cell/%s" % (tag, name)]
            j, f = t[..., n:2 * n].copy(), t[..., 2 * n:3 * n].copy()
            t[..., n:2 * n], t[..., 2 * n:3 * n] = f, j   # swap the candidate and forget blocks
    _, got = lstm.lstm_step(blob, gold["vec_main"], gold["state_main"], 32, 2, 0.5, 0.5, mode=0)
    assert np.abs(got - gold["out_main_qp32"]).max() > 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["main", "clip"])
def test_hip_cell_matches_torch_lstmcell(pkg, gold, which):
    c = pkg.EthCnn(device=0)
    c.load_lstm_blob(_blob(32))
    _, got = c.lstm_step(gold["vec_" + which], gold["state_" + which], 32, 2)
    c.close()
    want = gold["out_%s_qp32" % which]
    assert np.abs(got - want).max() <= TOL

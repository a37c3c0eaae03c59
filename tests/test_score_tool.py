"""CPU: tools/score_cu_depth.py (label / accuracy tooling): the vectorized score equals a
literal per-sample restatement of the reference's get_class_matrices (train_CNN_CTU64.py:103-137), and the
label reader accepts the reference's own AI_Info files when they are present."""
import glob
import importlib.util
import os

import numpy as np
import pytest

from conftest import have_reference

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("score_cu_depth", os.path.join(ROOT, "tools", "score_cu_depth.py"))
sc = importlib.util.module_from_spec(spec)
spec.loader.exec_module(sc)


def _literal(y_truth, p64, p32, p16, thr):
    m64, m32, m16 = [[0, 0], [0, 0]], [[0, 0], [0, 0]], [[0, 0], [0, 0]]
    idx = [[0, 1, 4, 5], [2, 3, 6, 7], [8, 9, 12, 13], [10, 11, 14, 15]]
    for i in range(y_truth.shape[0]):
        t = int(np.mean(y_truth[i]) > 0.5)
        m64[t][int(p64[i] > thr[0])] += 1
        if t:
            for j in range(4):
                t2 = int(np.mean(y_truth[i][idx[j]]) > 1.5)
                m32[t2][int(p32[i][j] > thr[1])] += 1
                if t2:
                    for k in range(4):
                        m16[int(y_truth[i][idx[j][k]] > 2.5)][int(p16[i][idx[j][k]] > thr[2])] += 1
    return [m64, m32, m16]


def test_vectorized_score_equals_literal():
    rng = np.random.default_rng(0)
    n = 500
    depth = rng.integers(0, 4, size=(n, 16))
    depth[:100] = 0
    depth[100:150] = 3
    probs = rng.random((n, 21)).astype(np.float32)
    for thr in ((0.5, 0.5, 0.5), (0.4, 0.3, 0.2), (0.6, 0.7, 0.8)):
        assert sc.class_matrices(depth, probs, thr) == _literal(depth, probs[:, 0], probs[:, 1:5], probs[:, 5:], thr)
    # perfect predictions built from the labels score 1.0 at every level
    perfect = np.zeros((n, 21), dtype=np.float32)
    perfect[:, 0] = depth.mean(axis=1) > 0.5
    perfect[:, 1:5] = depth[:, sc.IDX32].mean(axis=2) > 1.5
    perfect[:, 5:] = depth > 2.5
    assert [sc.accuracy(m) for m in sc.class_matrices(depth, perfect)] == [1.0, 1.0, 1.0]


def test_label_layout_round_trip(tmp_path):
    rng = np.random.default_rng(1)
    w, h, frames = 128, 192, 3
    lab = rng.integers(0, 4, size=(frames, h // 16, w // 16), dtype=np.uint8)
    p = tmp_path / "Info_x_CUDepth.dat"
    lab.tofile(str(p))
    back = sc.read_labels(str(p), w, h)
    assert np.array_equal(back, lab)
    per_ctu = sc.labels_per_ctu(back)
    assert per_ctu.shape == (frames * 2 * 3, 16)
    # CTU (frame 1, row 2, col 1): its 4x4 block map, raster
    assert np.array_equal(per_ctu[1 * 6 + 2 * 2 + 1].reshape(4, 4), lab[1, 8:12, 4:8])


@pytest.mark.skipif(not have_reference(), reason="needs /root/reference/AI_Info")
def test_reads_the_reference_label_files():
    files = sorted(glob.glob("/root/reference/AI_Info/*_768x512_*_CUDepth.dat"))
    assert files
    for f in files[:4]:
        lab = sc.read_labels(f, 768, 512)
        assert lab.shape[1:] == (32, 48) and lab.shape[0] == 50  # "..._nf50": 50 frames of (512/16) x (768/16) bytes
        assert sc.labels_per_ctu(lab).shape == (50 * 96, 16)

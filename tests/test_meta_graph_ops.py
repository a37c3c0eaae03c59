"""CPU: the per-op arithmetic of tests/meta_graph.py (the numpy interpreter that executes the reference's serialized
TensorFlow graphs) against PyTorch's own CPU kernels on the shapes, strides and paddings the graphs use.  The graph
WIRING is the reference's; this pins the one thing that is ours -- what each op computes -- to an independent
implementation (torch.nn.functional), op by op: Conv2D (VALID, stride = kernel), AvgPool (exact tiling),
ResizeNearestNeighbor, MatMul, Sigmoid, Maximum-as-leaky-ReLU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import meta_graph as mg


@pytest.mark.parametrize("hw,k,ci,co", [(64, 4, 1, 16), (16, 2, 16, 24), (8, 2, 24, 32), (32, 4, 1, 16), (4, 2, 16, 24), (16, 16, 1, 1)])
def test_conv2d_matches_torch(hw, k, ci, co):
    rng = np.random.default_rng(hw * 131 + k)
    x = rng.standard_normal((3, hw, hw, ci)).astype(np.float32)
    w = rng.standard_normal((k, k, ci, co)).astype(np.float32)
    got = mg._conv2d(x, w, [1, k, k, 1], "VALID")
    want = F.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(w).permute(3, 2, 0, 1).contiguous(), stride=k)
    want = want.permute(0, 2, 3, 1).numpy()
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("hw,k", [(64, 2), (64, 4)])
def test_avgpool_matches_torch(hw, k):
    x = np.random.default_rng(k).integers(0, 256, size=(2, hw, hw, 1)).astype(np.float32) / np.float32(255)
    got = mg._avgpool(x, [1, k, k, 1], [1, k, k, 1], "SAME")
    want = F.avg_pool2d(torch.from_numpy(x).permute(0, 3, 1, 2), k).permute(0, 2, 3, 1).numpy()
    assert np.abs(got - want).max() <= 1e-6


@pytest.mark.parametrize("src,dst", [(1, 16), (2, 32), (4, 64)])
def test_resize_nearest_matches_torch(src, dst):
    x = np.random.default_rng(src).standard_normal((2, src, src, 1)).astype(np.float32)
    got = mg._resize_nn(x, [dst, dst], False)
    want = F.interpolate(torch.from_numpy(x).permute(0, 3, 1, 2), size=(dst, dst), mode="nearest").permute(0, 2, 3, 1).numpy()
    assert np.array_equal(got, want)
    # and it is the block-mean broadcast the reference builds with it (net_CNN.py:78-84): every 16x16 block one value
    assert np.array_equal(got.reshape(2, src, 16, src, 16, 1)[:, :, 0, :, 0], x)


def test_matmul_sigmoid_maximum_match_torch():
    rng = np.random.default_rng(7)
    a = rng.standard_normal((5, 2688)).astype(np.float32)
    b = (rng.standard_normal((2688, 256)) / 50).astype(np.float32)
    it = mg.Interpreter({"m": {"op": "MatMul", "attr": {}, "inputs": ["a", "b"]}, "a": {"op": "Placeholder", "attr": {}, "inputs": []},
                         "b": {"op": "Placeholder", "attr": {}, "inputs": []},
                         "s": {"op": "Sigmoid", "attr": {}, "inputs": ["m"]},
                         "al": {"op": "Placeholder", "attr": {}, "inputs": []},
                         "mul": {"op": "Mul", "attr": {}, "inputs": ["al", "m"]},
                         "mx": {"op": "Maximum", "attr": {}, "inputs": ["mul", "m"]}}, {})
    m, s, mx = it.run(["m", "s", "mx"], {"a": a, "b": b, "al": np.float32(0.2)})
    tm = torch.from_numpy(a) @ torch.from_numpy(b)
    assert np.abs(m - tm.numpy()).max() <= 2e-5
    assert np.abs(s - torch.sigmoid(torch.from_numpy(m)).numpy()).max() <= 1e-6
    assert np.array_equal(mx, F.leaky_relu(torch.from_numpy(m), 0.2).numpy()) or \
        np.abs(mx - F.leaky_relu(torch.from_numpy(m), 0.2).numpy()).max() <= 1e-7

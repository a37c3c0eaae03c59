"""TEST INFRASTRUCTURE: a minimal interpreter for the reference's own serialized TensorFlow graph.

The reference ships `<model>.dat.meta` files: the MetaGraphDef TensorFlow 1.4.1 wrote from the authors'
graph.  This module walks that protobuf (no TensorFlow, no schema: the generic wire-format reader of
tests/golden/gen_meta_constants.py) and EXECUTES the inference subgraph node by node with numpy --
wiring, constants, strides, paddings, concat order, variable names all come from the reference's
file; only the per-op arithmetic (the documented semantics of ~20 TF ops) is supplied here.  Every
op output is rounded to float32 (as TF would); sums inside Conv2D / MatMul / AvgPool accumulate in
float64 (TF's fp32 summation order is unspecified; any order is within ~1e-6 of this).

Used (in the build container, where /root/reference exists) to generate tests/golden/
meta_exec_golden.npz and to check the oracle live; the committed vectors travel to the GPU box.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import gen_meta_constants as pb  # noqa: E402  (wire-format helpers)

DEAD = object()  # the untaken side of a Switch


def load_nodes(meta_path):
    raw = open(meta_path, "rb").read()
    graph = None
    for f, w, v in pb.fields(raw):
        if f == 2:
            graph = v
    nodes = {}
    for f, w, v in pb.fields(graph):
        if f != 1:
            continue
        node = {"op": "", "inputs": [], "attr": {}}
        name = ""
        for f2, w2, v2 in pb.fields(v):
            if f2 == 1:
                name = v2.decode()
            elif f2 == 2:
                node["op"] = v2.decode()
            elif f2 == 3:
                node["inputs"].append(v2.decode())
            elif f2 == 5:
                k = val = None
                for f3, w3, v3 in pb.fields(v2):
                    if f3 == 1:
                        k = v3.decode()
                    elif f3 == 2:
                        val = pb.parse_attr(v3)
                node["attr"][k] = val
        nodes[name] = node
    return nodes


def _const(node):
    t = node["attr"]["value"]["tensor"]
    shape = [int(d) for d in t["shape"]]
    if t["dtype"] == 1:
        vals, dt = t["floats"], np.float32
    elif t["dtype"] in (3, 9):
        vals, dt = t["ints"], np.int64
    else:
        raise NotImplementedError("Const dtype %r" % t["dtype"])
    a = np.array(vals if vals is not None else [], dtype=dt)
    n = int(np.prod(shape)) if shape else 1
    if a.size == 1 and n > 1:
        a = np.full(n, a[0], dtype=dt)  # splat encoding
    return a.reshape(shape)


def _f32(x):
    return np.asarray(x, dtype=np.float32)


def _conv2d(x, w, strides, padding):
    assert padding == "VALID" and strides[0] == strides[3] == 1
    n, h, wd, ci = x.shape
    kh, kw, ci2, co = w.shape
    assert ci == ci2
    sy, sx = strides[1], strides[2]
    oh, ow = (h - kh) // sy + 1, (wd - kw) // sx + 1
    out = np.zeros((n, oh, ow, co), dtype=np.float64)
    xd, wdb = x.astype(np.float64), w.astype(np.float64)
    for ky in range(kh):
        for kx in range(kw):
            patch = xd[:, ky:ky + sy * (oh - 1) + 1:sy, kx:kx + sx * (ow - 1) + 1:sx, :]
            out += patch @ wdb[ky, kx]
    return _f32(out)


def _avgpool(x, ksize, strides, padding):
    ky, kx, sy, sx = ksize[1], ksize[2], strides[1], strides[2]
    n, h, w, c = x.shape
    assert ky == sy and kx == sx and h % ky == 0 and w % kx == 0, "only exact tilings occur in the reference graph"
    return _f32(x.astype(np.float64).reshape(n, h // ky, ky, w // kx, kx, c).mean(axis=(2, 4)))


def _resize_nn(x, size, align_corners):
    assert not align_corners
    n, h, w, c = x.shape
    oh, ow = int(size[0]), int(size[1])
    ys = np.minimum((np.arange(oh) * (h / oh)).astype(np.int64), h - 1)  # floor(y * scale)
    xs = np.minimum((np.arange(ow) * (w / ow)).astype(np.int64), w - 1)
    return x[:, ys][:, :, xs]


class Interpreter(object):
    def __init__(self, nodes, variables):
        self.nodes, self.vars = nodes, variables
        self.feeds, self.memo = {}, {}
        self.ops_used = set()

    def run(self, fetches, feeds):
        self.feeds, self.memo = feeds, {}
        return [self.value(f) for f in fetches]

    def value(self, ref):
        if ref.startswith("^"):
            return None  # control edge
        name, _, port = ref.partition(":")
        port = int(port) if port else 0
        key = (name, port)
        if key not in self.memo:
            outs = self.eval_node(name)
            for i, o in enumerate(outs):
                self.memo[(name, i)] = o
        return self.memo[key]

    def eval_node(self, name):
        nd = self.nodes[name]
        op, attr = nd["op"], nd["attr"]
        if op == "Placeholder":
            return [self.feeds[name]]
        if op == "Const":
            return [_const(nd)]
        if op == "VariableV2":
            return [_f32(self.vars[name])]
        if op == "Merge":  # lazy: exactly one input is alive
            alive = [(i, v) for i, v in enumerate(self.value(r) for r in nd["inputs"]) if v is not DEAD]
            assert len(alive) == 1, "%s: %d live inputs" % (name, len(alive))
            return [alive[0][1], np.int32(alive[0][0])]
        ins = [self.value(r) for r in nd["inputs"] if not r.startswith("^")]
        if op == "Switch":
            data, pred = ins
            if data is DEAD or pred is DEAD:
                return [DEAD, DEAD]
            return [DEAD, data] if bool(pred) else [data, DEAD]
        if any(v is DEAD for v in ins):
            return [DEAD]
        self.ops_used.add(op)  # ops that really executed (the dropout branches stay dead)
        if op == "Identity":
            return [ins[0]]
        if op == "Mul":
            return [_f32(_f32(ins[0]) * _f32(ins[1]))]
        if op == "Add":
            return [_f32(_f32(ins[0]) + _f32(ins[1]))]
        if op == "Sub":
            return [_f32(_f32(ins[0]) - _f32(ins[1]))]
        if op == "RealDiv":
            return [_f32(_f32(ins[0]) / _f32(ins[1]))]
        if op == "Maximum":
            return [np.maximum(_f32(ins[0]), _f32(ins[1]))]
        if op == "Relu":
            return [np.maximum(_f32(ins[0]), np.float32(0))]
        if op == "Tanh":
            return [_f32(np.tanh(ins[0].astype(np.float64)))]
        if op == "Sigmoid":
            return [_f32(1.0 / (1.0 + np.exp(-ins[0].astype(np.float64))))]
        if op == "Less":
            return [np.asarray(ins[0] < ins[1])]
        if op == "Greater":
            return [np.asarray(ins[0] > ins[1])]
        if op == "Reshape":
            return [np.reshape(ins[0], [int(d) for d in ins[1]])]
        if op == "ConcatV2":
            return [np.concatenate(ins[:-1], axis=int(ins[-1]))]
        if op == "MatMul":
            a, b = ins
            if attr.get("transpose_a", {}).get("b"):
                a = a.T
            if attr.get("transpose_b", {}).get("b"):
                b = b.T
            return [_f32(a.astype(np.float64) @ b.astype(np.float64))]
        if op == "Conv2D":
            return [_conv2d(ins[0], ins[1], attr["strides"]["list_i"], attr["padding"]["s"])]
        if op == "AvgPool":
            return [_avgpool(ins[0], attr["ksize"]["list_i"], attr["strides"]["list_i"], attr["padding"]["s"])]
        if op == "ResizeNearestNeighbor":
            return [_resize_nn(ins[0], ins[1], attr.get("align_corners", {}).get("b", False))]
        raise NotImplementedError("op %s (node %s)" % (op, name))


# ---- the two graphs the reference ships -------------------------------------------------------
AI_META = "/root/reference/HM-16.5_Test_AI/bin/model_2000000_qp30~35.dat.meta"
LDP_CNN_META = "/root/reference/HM-16.5_Test_LDP/bin/model_LDP_2000000_qp22~37.dat.meta"
# AI graph (the training script's graph: same layers as net_CNN.py:103-185, NO threshold gates --
# those exist only in the test-time net_CNN.py:175,187): feeds and fetches by node name
AI_FEEDS = {"x": "Placeholder", "qp": "Placeholder_2", "isdrop": "Placeholder_3"}
AI_FETCHES = ["cond_2/Merge", "cond_5/Merge", "cond_8/Merge", "concat"]  # y64, y32, y16 (ungated), h_conv_flat


def run_ai_graph(nodes, variables, ctus_u8, qp):
    """ctus_u8 [n,64,64] -> (probs [n,21] ungated, features [n,2688]) through the reference's graph"""
    n = ctus_u8.shape[0]
    it = Interpreter(nodes, variables)
    feeds = {AI_FEEDS["x"]: ctus_u8.reshape(n, 64, 64, 1).astype(np.float32),
             AI_FEEDS["qp"]: np.full((n, 1), float(qp), dtype=np.float32),
             AI_FEEDS["isdrop"]: np.float32(0.0)}
    y64, y32, y16, feat = it.run(AI_FETCHES, feeds)
    return np.concatenate([y64, y32, y16], axis=1), feat, it.ops_used


def _consumers(nodes, name):
    return [n for n, nd in nodes.items() if any(r.lstrip("^").split(":")[0] == name for r in nd["inputs"])
            and not n.startswith(("gradients", "save", "Adam", "Momentum"))]


def fc1_output_nodes(nodes):
    """h_fc1_{64,32,16} = LeakyRelu(MatMul(h_conv_flat, h_fc1__XX__w) + b): found by following the
    variable's consumers (the LDP graph adds batch-norm-free FC blocks with different node numbers)"""
    outs = []
    for tag in ("64", "32", "16"):
        read = "h_fc1__%s__w/read" % tag
        mm = [n for n in _consumers(nodes, read) if nodes[n]["op"] == "MatMul"]
        assert len(mm) == 1, mm
        add = [n for n in _consumers(nodes, mm[0]) if nodes[n]["op"] == "Add"]
        assert len(add) == 1, add
        mx = [n for n in _consumers(nodes, add[0]) if nodes[n]["op"] == "Maximum"]
        assert len(mx) == 1, mx
        outs.append(mx[0])
    return outs


def run_resi_graph(nodes, variables, ctus_u8):
    """LDP CNN graph (resi_cnn, net_CNN_LSTM_one_step.py:151-199 = the pre-training graph the authors
    saved): residual CTUs [n,64,64] u8 -> the 448-vector [h_fc1_64 | h_fc1_32 | h_fc1_16]"""
    n = ctus_u8.shape[0]
    it = Interpreter(nodes, variables)
    feeds = {"Placeholder": ctus_u8.reshape(n, 64, 64, 1).astype(np.float32)}
    outs = it.run(fc1_output_nodes(nodes), feeds)
    return np.concatenate(outs, axis=1), it.ops_used

"""TEST INFRASTRUCTURE: a numpy stand-in for the slice of TensorFlow 1.x that the reference's four hot-path
Python files touch, so that THOSE FILES THEMSELVES can be executed in the build container (where
/root/reference exists and TensorFlow does not):

    HM-16.5_Test_AI/bin/video_to_cu_depth.py   (run as __main__ with its real argv)
    HM-16.5_Test_AI/bin/net_CNN.py
    HM-16.5_Test_LDP/bin/resi_to_cu_depth_LDP.py   (its __main__ daemon loop, driven over its file protocol)
    HM-16.5_Test_LDP/bin/net_CNN_LSTM_one_step.py

What this pins and what it does not.  Every line of Python between `import tensorflow as tf` and the bytes
of cu_depth.dat / state.dat is the reference's own: reading and zero-padding the frame, the tiling loop, the
1024-CTU sub-batching, the feed_dict, the layer wiring, concat orders, the qp / efs columns, the two
batch-level tf.cond gates, the QP-band model switch, the LSTM state slicing, the daemon's command parsing.
What is NOT the reference's is the arithmetic inside each `tf.*` call: the ~30 functions below restate the
documented semantics of the corresponding TF 1.x ops (each op output rounded to float32 as TF does; sums
inside Conv2D / MatMul / AvgPool accumulate in float64 because TF's fp32 summation order is unspecified --
any order is within ~1e-6 of this).  So outputs produced through this module pin the oracle and the HIP path
to "the reference's program over stand-in op kernels", not to a TensorFlow run; oracle/ethcnn_oracle.c keeps
saying "parity unpinned" for exactly that remainder.

Graphs are lazy (like TF 1.x): `tf.*` calls build nodes, `Session.run` evaluates only what the fetches
reach (so the label plumbing of net_CNN.py:108-121 and the dropout branches never run, as in TF), and
`tf.cond` evaluates only the taken branch.  Variables hold no value until `Saver.restore` reads them BY NAME
from a TF-V2 bundle (<prefix>.index + <prefix>.data-00000-of-00001) in the working directory, as TF would.

Only tests/golden/gen_ref_exec_golden.py and tests/test_ref_exec.py import this; nothing in the product does.
"""
import collections
import contextlib
import os
import struct
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)
from meta_graph import _avgpool, _conv2d, _f32, _resize_nn  # noqa: E402  (the same op kernels the .meta interpreter uses)

# TF_SHIM_KERNELS=torch: the ops whose arithmetic is more than one rounding per element (Conv2D, AvgPool, MatMul,
# Sigmoid, Tanh) and ResizeNearestNeighbor are computed by PYTORCH'S OWN CPU KERNELS in float32 (its accumulation
# order, its exp / tanh) instead of the numpy restatements: the reference's program over a third party's op
# kernels, nothing of the arithmetic written here.  Elementwise one-rounding ops (Add, Mul, Maximum ...) stay numpy:
# IEEE leaves them no freedom.  Default "numpy" (float64 accumulation, rounded once) is what the committed fixture
# tests/golden/ref_exec_golden.npz was produced with; ref_exec_golden_torch.npz holds the torch-kernel outputs.
KERNELS = os.environ.get("TF_SHIM_KERNELS", "numpy")
if KERNELS not in ("numpy", "torch"):
    raise ValueError("TF_SHIM_KERNELS must be numpy or torch")
if KERNELS == "torch":
    import torch
    import torch.nn.functional as _F

    def _t(a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))

    def _conv2d(x, w, strides, padding):  # noqa: F811  NHWC x HWIO, VALID (the only padding the reference's convs use)
        assert padding == "VALID" and strides[0] == 1 and strides[3] == 1, (strides, padding)
        y = _F.conv2d(_t(x).permute(0, 3, 1, 2), _t(w).permute(3, 2, 0, 1).contiguous(), stride=(strides[1], strides[2]))
        return y.permute(0, 2, 3, 1).contiguous().numpy()

    def _avgpool(x, ksize, strides, padding):  # noqa: F811  SAME with an exact tiling == no padding
        kh, kw = ksize[1], ksize[2]
        assert list(ksize) == list(strides) and x.shape[1] % kh == 0 and x.shape[2] % kw == 0, (ksize, strides, x.shape)
        return _F.avg_pool2d(_t(x).permute(0, 3, 1, 2), (kh, kw)).permute(0, 2, 3, 1).contiguous().numpy()

    def _resize_nn(x, size, align_corners):  # noqa: F811
        assert not align_corners
        y = _F.interpolate(_t(x).permute(0, 3, 1, 2), size=(int(size[0]), int(size[1])), mode="nearest")
        return y.permute(0, 2, 3, 1).contiguous().numpy()

float32 = np.float32
int32 = np.int32
# tf.bool is bound at the bottom of the file (the name shadows the builtin, which nothing below needs)


# ----------------------------------------------------------------------------------------------- graph ----
class _Graph(object):
    def __init__(self):
        self.names = collections.Counter()
        self.scope = []           # variable_scope stack
        self.trainable = []       # creation order (tf.trainable_variables)

    def unique(self, base):
        k = self.names[base]
        self.names[base] += 1
        return base if k == 0 else "%s_%d" % (base, k)


_G = _Graph()
OPS_USED = set()  # names of the stand-in ops that really executed in some Session.run (reported in the golden file)


def reset_default_graph():
    global _G
    _G = _Graph()


class Tensor(object):
    """a node: fn(*evaluated inputs) -> ndarray.  Operators follow TF's: Python scalars convert to float32."""
    __array_priority__ = 1000

    def __init__(self, op, fn, inputs, name=None):
        self.op, self.fn, self.inputs = op, fn, list(inputs)
        self.name = name or _G.unique(op)

    def __repr__(self):
        return "<tf_shim.Tensor '%s:0' op=%s>" % (self.name, self.op)

    __hash__ = object.__hash__

    def __add__(self, o): return _binary("Add", np.add, self, o)
    def __radd__(self, o): return _binary("Add", np.add, o, self)
    def __sub__(self, o): return _binary("Sub", np.subtract, self, o)
    def __rsub__(self, o): return _binary("Sub", np.subtract, o, self)
    def __mul__(self, o): return _binary("Mul", np.multiply, self, o)
    def __rmul__(self, o): return _binary("Mul", np.multiply, o, self)
    def __truediv__(self, o): return _binary("RealDiv", np.divide, self, o)
    def __lt__(self, o): return _binary("Less", np.less, self, o, out_float=False)
    def __gt__(self, o): return _binary("Greater", np.greater, self, o, out_float=False)

    def __getitem__(self, idx):  # StridedSlice
        return Tensor("StridedSlice", lambda a: a[idx], [self])


def _const_tensor(v, dtype=None, name="Const"):
    a = np.asarray(v, dtype=dtype)
    return Tensor("Const", lambda: a, [], _G.unique(name))


def _lift(v, like_float=True):
    """ops.convert_to_tensor: tensors pass, lists of tensors pack (tf.stack axis 0), Python numbers become
    float32 constants when the other operand is a float tensor (TF converts to the tensor operand's dtype)"""
    if isinstance(v, Tensor):
        return v
    if isinstance(v, (list, tuple)) and any(isinstance(e, Tensor) for e in v):
        return stack(list(v), 0)
    if isinstance(v, (np.ndarray, np.generic)):
        return _const_tensor(v)
    return _const_tensor(v, np.float32 if like_float else None)


def _binary(op, ufunc, a, b, out_float=True):
    a, b = _lift(a), _lift(b)

    def fn(x, y):
        OPS_USED.add(op)
        if x.dtype.kind == "f" or y.dtype.kind == "f":
            x, y = _f32(x), _f32(y)
        r = ufunc(x, y)
        return _f32(r) if (out_float and r.dtype.kind == "f") else r
    return Tensor(op, fn, [a, b])


def _unary(op, f, a):
    def fn(x):
        OPS_USED.add(op)
        return f(x)
    return Tensor(op, fn, [_lift(a)])


class _Cond(Tensor):
    pass


class Session(object):
    def __init__(self, *a, **k):
        self.values = {}  # Variable -> ndarray

    def run(self, fetches, feed_dict=None):
        single = not isinstance(fetches, (list, tuple))
        memo = {}
        for ph, v in (feed_dict or {}).items():
            assert ph.op == "Placeholder", ph
            memo[ph] = np.asarray(v, dtype=ph.dtype)  # TF casts a fed value to the placeholder's dtype
        out = [self._eval(f, memo) for f in ([fetches] if single else fetches)]
        return out[0] if single else out

    def _eval(self, t, memo):
        # iterative post-order walk (the graphs are shallow, but recursion limits are not ours to spend)
        stack_ = [t]
        while stack_:
            n = stack_[-1]
            if n in memo:
                stack_.pop()
                continue
            if n.op == "Placeholder":
                raise RuntimeError("You must feed a value for placeholder tensor %r" % n.name)
            if isinstance(n, Variable):
                if n not in self.values:
                    raise RuntimeError("Attempting to use uninitialized value " + n.name)
                memo[n] = self.values[n]
                stack_.pop()
                continue
            if isinstance(n, _Cond):
                pred = n.inputs[0]
                if pred not in memo:
                    stack_.append(pred)
                    continue
                OPS_USED.add("cond")
                taken = n.inputs[1] if builtins_bool(memo[pred]) else n.inputs[2]
                if taken not in memo:
                    stack_.append(taken)
                    continue
                memo[n] = memo[taken]
                stack_.pop()
                continue
            missing = [i for i in n.inputs if i not in memo]
            if missing:
                stack_.extend(missing)
                continue
            memo[n] = n.fn(*[memo[i] for i in n.inputs])
            stack_.pop()
        return memo[t]


def builtins_bool(v):
    return True if np.asarray(v).reshape(-1)[0] else False


# ----------------------------------------------------------------------------------- variables, scopes ----
class Variable(Tensor):
    def __init__(self, initial_value=None, name=None, shape=None, _full_name=None):
        prefix = "".join(s + "/" for s in _G.scope)
        nm = _full_name or _G.unique(prefix + (name or "Variable"))
        Tensor.__init__(self, "VariableV2", None, [], nm)
        self.shape = tuple(int(d) for d in (shape if shape is not None else initial_value.static_shape))
        _G.trainable.append(self)


def get_variable(name, shape=None, **k):
    full = "".join(s + "/" for s in _G.scope) + name
    for v in _G.trainable:
        if v.name == full:
            raise ValueError("Variable %s already exists, disallowed (reuse is not set)" % full)
    _G.names[full] += 1
    return Variable(shape=shape, _full_name=full)


def trainable_variables():
    return list(_G.trainable)


class _Scope(object):
    def reuse_variables(self):
        raise NotImplementedError("reuse_variables: LSTM_READ_LENGTH is 1 in the reference; never reached")


@contextlib.contextmanager
def variable_scope(name, *a, **k):
    _G.scope.append(name)
    try:
        yield _Scope()
    finally:
        _G.scope.pop()


def get_variable_scope():
    return _Scope()


# ---------------------------------------------------------------------- TF-V2 bundle reader (by name) ----
def _varint(b, i):
    v = s = 0
    while True:
        c = b[i]
        i += 1
        v |= (c & 0x7F) << s
        s += 7
        if not c & 0x80:
            return v, i


def _read_block(raw, off, size):
    blk = raw[off:off + size]
    nrestart = struct.unpack("<I", blk[-4:])[0]
    end = len(blk) - 4 - 4 * nrestart
    i, key, out = 0, b"", []
    while i < end:
        shared, i = _varint(blk, i)
        non_shared, i = _varint(blk, i)
        vlen, i = _varint(blk, i)
        key = key[:shared] + blk[i:i + non_shared]
        i += non_shared
        out.append((key, blk[i:i + vlen]))
        i += vlen
    return out


def read_bundle_index(index_path):
    """name -> (shape, offset, size): the LevelDB-style table TF's BundleWriter emits"""
    raw = open(index_path, "rb").read()
    footer = raw[-48:]
    assert struct.unpack("<Q", footer[40:])[0] == 0xDB4775248B80FB57, "not a TF-V2 .index table"
    _, i = _varint(footer, 0)
    _, i = _varint(footer, i)
    idx_off, i = _varint(footer, i)
    idx_size, i = _varint(footer, i)
    entries = {}
    for _, handle in _read_block(raw, idx_off, idx_size):
        boff, j = _varint(handle, 0)
        bsize, j = _varint(handle, j)
        for key, val in _read_block(raw, boff, bsize):
            if key == b"":
                continue  # BundleHeaderProto
            shape, offset, size, i2 = [], 0, 0, 0
            while i2 < len(val):
                tag, i2 = _varint(val, i2)
                f, w = tag >> 3, tag & 7
                if w == 0:
                    v, i2 = _varint(val, i2)
                    if f == 4:
                        offset = v
                    elif f == 5:
                        size = v
                elif w == 2:
                    ln, i2 = _varint(val, i2)
                    sub, i2 = val[i2:i2 + ln], i2 + ln
                    if f == 2:  # TensorShapeProto{ repeated Dim dim = 2 { int64 size = 1 } }
                        k = 0
                        while k < len(sub):
                            t2, k = _varint(sub, k)
                            l2, k = _varint(sub, k)
                            d, k = sub[k:k + l2], k + l2
                            if t2 >> 3 == 2:
                                dv, _ = _varint(d, 1) if d else (0, 0)
                                shape.append(dv)
                elif w == 5:
                    i2 += 4
                else:
                    raise ValueError("wire type %d in BundleEntryProto" % w)
            entries[key.decode()] = (tuple(shape), offset, size)
    return entries


class _SaverDef(object):
    V1, V2 = 1, 2


class Saver(object):
    def __init__(self, var_list=None, write_version=None, **k):
        self.var_list = list(var_list if var_list is not None else _G.trainable)

    def restore(self, sess, save_path):
        index = read_bundle_index(save_path + ".index")
        data = open(save_path + ".data-00000-of-00001", "rb").read()
        for v in self.var_list:
            if v.name not in index:
                raise KeyError("Key %s not found in checkpoint %s" % (v.name, save_path))
            shape, off, size = index[v.name]
            if tuple(shape) != v.shape:
                raise ValueError("restore %s: checkpoint shape %r, variable shape %r" % (v.name, shape, v.shape))
            assert size == 4 * int(np.prod(shape)) and off + size <= len(data)
            sess.values[v] = np.frombuffer(data, dtype="<f4", count=size // 4, offset=off).reshape(shape).copy()
        RESTORED.append((os.path.basename(save_path), len(self.var_list)))


RESTORED = []  # (file name, number of variables) per Saver.restore call: the model-switch evidence


class _Train(object):
    Saver = Saver
    SaverDef = _SaverDef


train = _Train()


# ------------------------------------------------------------------------------------------------ ops ----
class _Init(object):
    """the value of tf.truncated_normal(...): only its static shape is ever used (every variable is restored)"""

    def __init__(self, shape):
        self.static_shape = tuple(int(d) for d in shape)


def truncated_normal(shape, stddev=1.0, **k):
    return _Init(shape)


def placeholder(dtype, shape=None, name=None):
    t = Tensor("Placeholder", None, [], _G.unique(name or "Placeholder"))
    t.dtype = np.float32 if dtype in ("float", np.float32, "float32") else np.dtype(dtype)
    return t


def constant(value, dtype=None, shape=None, name="Const"):
    a = np.asarray(value, dtype=np.float32 if dtype in (None, np.float32) else dtype)
    if shape is not None:
        a = np.full([int(d) for d in shape], a.reshape(-1)[0], dtype=a.dtype) if a.size == 1 else a.reshape(shape)
    return _const_tensor(a, name=name)


def cast(x, dtype):
    return _unary("Cast", lambda a: np.asarray(a).astype(dtype), _lift(x, like_float=False))


def to_int32(x):
    return _unary("Cast", lambda a: a.astype(np.int32), x)  # float -> int32 truncates toward zero, as numpy's astype


def scalar_mul(scalar, x):
    return _binary("Mul", np.multiply, scalar, x)


def multiply(a, b):
    return _binary("Mul", np.multiply, a, b)


def reshape(x, shape):
    parts = [(_lift(d, like_float=False) if isinstance(d, Tensor) else None) for d in shape]
    dyn = [p for p in parts if p is not None]

    def fn(a, *dv):
        OPS_USED.add("Reshape")
        it = iter(dv)
        return np.reshape(a, [int(next(it)) if p is not None else int(d) for p, d in zip(parts, shape)])
    return Tensor("Reshape", fn, [_lift(x)] + dyn)


def concat(values, axis, name=None):
    def fn(*a):
        OPS_USED.add("ConcatV2")
        return np.concatenate(a, axis=axis)
    return Tensor("ConcatV2", fn, [_lift(v) for v in values])


def stack(values, axis=0):
    def fn(*a):
        OPS_USED.add("Pack")
        return np.stack(a, axis=axis)
    return Tensor("Pack", fn, [_lift(v) for v in values])


def slice(x, begin, size):  # noqa: A001
    def fn(a):
        OPS_USED.add("Slice")
        idx = tuple(np.s_[b:(None if s == -1 else b + s)] for b, s in zip(begin, size))
        return a[idx]
    return Tensor("Slice", fn, [_lift(x)])


def split(x, num_or_size_splits, axis=0):
    n = int(num_or_size_splits)
    x = _lift(x)

    def part(k):
        def fn(a):
            OPS_USED.add("Split")
            return np.split(a, n, axis=axis)[k]
        return Tensor("Split", fn, [x])
    return [part(k) for k in range(n)]


def one_hot(indices, depth, axis=-1):
    def fn(a):
        OPS_USED.add("OneHot")
        oh = (a[..., None] == np.arange(depth)).astype(np.float32)  # on 1.0 / off 0.0, float32 (TF defaults)
        return oh if axis in (-1, a.ndim) else np.moveaxis(oh, -1, axis)
    return Tensor("OneHot", fn, [_lift(indices, like_float=False)])


def shape(x):  # noqa: A001
    return _unary("Shape", lambda a: np.asarray(a.shape, dtype=np.int32), x)


def zeros(shape, dtype=np.float32):  # noqa: A002
    parts = [d if isinstance(d, Tensor) else None for d in shape]
    dyn = [p for p in parts if p is not None]

    def fn(*dv):
        OPS_USED.add("Fill")
        it = iter(dv)
        return np.zeros([int(next(it)) if p is not None else int(d) for p, d in zip(parts, shape)], dtype=dtype)
    return Tensor("Fill", fn, dyn)


def count_nonzero(x):
    return _unary("count_nonzero", lambda a: np.int64(np.count_nonzero(a)), _lift(x, like_float=False))


def matmul(a, b):
    def fn(x, y):
        OPS_USED.add("MatMul")
        if KERNELS == "torch":
            return torch.matmul(_t(x), _t(y)).numpy()
        return _f32(x.astype(np.float64) @ y.astype(np.float64))
    return Tensor("MatMul", fn, [_lift(a), _lift(b)])


def cond(pred, true_fn=None, false_fn=None, fn1=None, fn2=None):
    """TF 1.x builds both branches at graph time and runs one; here both are built, one is evaluated"""
    t = _Cond("cond", None, [_lift(pred, like_float=False), _lift((true_fn or fn1)()), _lift((false_fn or fn2)())])
    return t


def _sigmoid(a):
    if KERNELS == "torch":
        return torch.sigmoid(_t(a)).numpy()
    return _f32(1.0 / (1.0 + np.exp(-a.astype(np.float64))))


def _tanh(a):
    if KERNELS == "torch":
        return torch.tanh(_t(a)).numpy()
    return _f32(np.tanh(a.astype(np.float64)))


class _NN(object):
    @staticmethod
    def conv2d(input, filter, strides, padding, **k):  # noqa: A002
        def fn(x, w):
            OPS_USED.add("Conv2D")
            return _conv2d(_f32(x), _f32(w), list(strides), padding)
        return Tensor("Conv2D", fn, [_lift(input), _lift(filter)])

    @staticmethod
    def avg_pool(value, ksize, strides, padding, **k):
        return _unary("AvgPool", lambda a: _avgpool(_f32(a), list(ksize), list(strides), padding), value)

    @staticmethod
    def relu(x):
        return _unary("Relu", lambda a: np.maximum(_f32(a), np.float32(0)), x)

    @staticmethod
    def leaky_relu(x, alpha=0.2):
        # TF emits Maximum(alpha * x, x) with alpha a float32 constant (tests/golden/meta_constants.json: 0.20000000298)
        al = np.float32(alpha)
        return _unary("LeakyRelu", lambda a: np.maximum(_f32(al * _f32(a)), _f32(a)), x)

    @staticmethod
    def sigmoid(x):
        return _unary("Sigmoid", _sigmoid, x)

    @staticmethod
    def dropout(x, keep_prob, **k):
        def fn(a, kp):
            raise RuntimeError("dropout executed: isdrop must be 0 at inference (net_CNN.py:99)")
        return Tensor("Dropout", fn, [_lift(x), _lift(keep_prob)])

    @staticmethod
    def max_pool(*a, **k):
        raise NotImplementedError("max_pool_2x2 is defined but never called by the reference")


nn = _NN()


class _Image(object):
    @staticmethod
    def resize_nearest_neighbor(images, size, align_corners=False):
        return _unary("ResizeNearestNeighbor", lambda a: _resize_nn(a, size, align_corners), images)


image = _Image()


def reduce_mean(*a, **k):
    raise NotImplementedError("zero_mean_norm_global is defined but never called by the reference")


tile = reduce_mean


# ------------------------------------------------------------------------------- tf.contrib.rnn subset ----
LSTMStateTuple = collections.namedtuple("LSTMStateTuple", ("c", "h"))


class LSTMCell(object):
    """tf.contrib.rnn.LSTMCell (rnn_cell_impl.LSTMCell.call, TF 1.x), no peepholes, no projection:
        z = [x, h_prev] . kernel + bias ;  i, j, f, o = split(z, 4, axis 1)
        c = sigmoid(f + forget_bias) * c_prev + sigmoid(i) * tanh(j) ;  c = clip(c, -cell_clip, cell_clip)
        h = sigmoid(o) * tanh(c) ;  returns h, LSTMStateTuple(c, h)
    variables `<scope>/lstm_cell/kernel` [in + units, 4 units] and `<scope>/lstm_cell/bias` [4 units]"""

    def __init__(self, num_units, forget_bias=1.0, cell_clip=None, state_is_tuple=True, **k):
        self.n, self.fb, self.clip = int(num_units), np.float32(forget_bias), cell_clip
        self.kernel = self.bias = None

    def __call__(self, inputs, state, input_depth=None):
        c_prev, h_prev = state
        with variable_scope("lstm_cell"):
            if self.kernel is None:
                # input depth: the reference feeds x of width num_units (NUM_VECTOR_SIZE == NUM_HIDDEN_SIZE)
                depth = self.n if input_depth is None else input_depth
                self.kernel = get_variable("kernel", [depth + self.n, 4 * self.n])
                self.bias = get_variable("bias", [4 * self.n])
        z = matmul(concat([inputs, h_prev], 1), self.kernel) + self.bias
        i, j, f, o = split(z, 4, axis=1)
        c = nn.sigmoid(f + self.fb) * c_prev + nn.sigmoid(i) * _unary("Tanh", _tanh, j)
        if self.clip is not None:
            lim = np.float32(self.clip)
            c = _unary("ClipByValue", lambda a: np.minimum(np.maximum(_f32(a), -lim), lim), c)
        h = nn.sigmoid(o) * _unary("Tanh", _tanh, c)
        return h, LSTMStateTuple(c, h)


class DropoutWrapper(object):
    """output_keep_prob is a tensor (1 - isdrop * 0.5), so TF always emits nn.dropout on the output:
    x / keep * floor(keep + U[0,1)).  With keep == 1 that is x / 1 * 1 == x bit for bit; any other keep is refused."""

    def __init__(self, cell, output_keep_prob=1.0, **k):
        self.cell, self.keep = cell, _lift(output_keep_prob)

    def __call__(self, inputs, state):
        out, new_state = self.cell(inputs, state)

        def fn(a, kp):
            OPS_USED.add("DropoutWrapper(keep=1)")
            if float(kp) != 1.0:
                raise RuntimeError("DropoutWrapper with keep_prob %r: isdrop must be 0 at inference" % float(kp))
            return a
        return Tensor("Dropout", fn, [out, self.keep]), new_state


class MultiRNNCell(object):
    def __init__(self, cells, state_is_tuple=True):
        self.cells = list(cells)

    def __call__(self, inputs, state):
        new_states, cur = [], inputs
        with variable_scope("multi_rnn_cell"):
            for i, cell in enumerate(self.cells):
                with variable_scope("cell_%d" % i):
                    cur, ns = cell(cur, state[i])
                    new_states.append(ns)
        return cur, tuple(new_states)


class _Rnn(object):
    LSTMCell = LSTMCell
    DropoutWrapper = DropoutWrapper
    MultiRNNCell = MultiRNNCell
    LSTMStateTuple = LSTMStateTuple


class _Contrib(object):
    rnn = _Rnn()


contrib = _Contrib()


bool = np.bool_  # noqa: A001  (tf.bool)


def install():
    """put this module where `import tensorflow` finds it (build container only)"""
    sys.modules["tensorflow"] = sys.modules[__name__]

"""ctypes binding of libethcnn.so (include/ethcnn.h) -- the only way Python reaches the
HIP kernels.  No torch, no numpy math: numpy is used for buffers only.

There is deliberately no fallback: if the shared library is missing, or no gfx950 device
is usable, construction raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libethcnn.so")
LIB_PATH = os.environ.get("ETHCNN_LIB", LIB_PATH)  # development knob: A/B builds of the same library

NOUT, NFEAT, NVEC, NFC2, SUB_BATCH = 21, 2688, 448, 336, 1024
BLOB_FLOATS = 1288210
LSTM_BLOB_FLOATS = 760078

STAGES = ("tile", "trunk", "fc1", "heads", "gate")
DBG_FEATURES, DBG_FC1, DBG_FC2, DBG_LOGITS, DBG_RAW_PROBS = range(5)
_DBG_WIDTH = {DBG_FEATURES: NFEAT, DBG_FC1: NVEC, DBG_FC2: NFC2, DBG_LOGITS: NOUT, DBG_RAW_PROBS: NOUT}


ERR_ROWS_TIMEOUT, ERR_PLAN_REFUSED = -7, -8  # include/ethcnn.h


def fast_plan_bound(blob, plan, lib=None):
    """a-priori floor bound of plan 2 / 3 on a probability for a blob in host memory (no device) -> (accepted_by_bound_alone, bound, feature_bound)"""
    lib = lib or load_library()
    blob = np.ascontiguousarray(blob, dtype=np.float32)
    b, f = ctypes.c_double(), ctypes.c_double()
    rc = lib.ethcnn_fast_plan_bound(blob.ctypes.data, blob.size, int(plan), ctypes.byref(b), ctypes.byref(f))
    if rc not in (0, ERR_PLAN_REFUSED):
        raise EthCnnError(rc, "ethcnn_fast_plan_bound: bad arguments")
    return rc == 0, b.value, f.value


class EthCnnError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "libethcnn error %d: %s" % (code, msg))
        self.code = code


class Options(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int), ("max_ctus_per_pass", ctypes.c_int), ("host_threads", ctypes.c_int),
                ("reserved", ctypes.c_int * 5)]


class StageTimes(ctypes.Structure):
    _fields_ = [("ms", ctypes.c_double * 5), ("launches", ctypes.c_int64 * 5), ("ctus", ctypes.c_int64),
                ("timed", ctypes.c_int64 * 5), ("timed_ctus", ctypes.c_int64 * 5), ("timing_errors", ctypes.c_int64)]


class CkptEntry(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 64), ("dtype", ctypes.c_int), ("rank", ctypes.c_int),
                ("shape", ctypes.c_int64 * 4), ("shard", ctypes.c_int), ("offset", ctypes.c_int64),
                ("size", ctypes.c_int64), ("crc32c", ctypes.c_uint32)]


# every symbol include/ethcnn.h declares: name -> (restype, argtypes)
_vp, _cp, _i, _sz = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_size_t
_fp, _pd = ctypes.POINTER(ctypes.c_float), ctypes.c_ssize_t
SIGNATURES = {
    "ethcnn_create": (_i, [ctypes.POINTER(_vp), ctypes.POINTER(Options)]),
    "ethcnn_destroy": (None, [_vp]),
    "ethcnn_last_error": (_cp, [_vp]),
    "ethcnn_version": (_cp, []),
    "ethcnn_load_checkpoint": (_i, [_vp, _cp]),
    "ethcnn_load_blob": (_i, [_vp, _fp, _sz]),
    "ethcnn_load_synthetic": (_i, [_vp, ctypes.c_uint64, ctypes.c_double]),
    "ethcnn_get_blob": (_i, [_vp, _fp, _sz]),
    "ethcnn_model_name_for_qp": (_i, [_i, ctypes.c_char_p, _sz]),
    "ethcnn_load_thresholds": (_i, [_vp, _cp]),
    "ethcnn_parse_thresholds": (_i, [_cp, _fp, _fp]),
    "ethcnn_set_thresholds": (_i, [_vp, ctypes.c_float, ctypes.c_float]),
    "ethcnn_get_thresholds": (_i, [_vp, _fp, _fp]),
    "ethcnn_predict_luma_device": (_i, [_vp, _vp, _i, _i, _pd, _pd, _i, _i, _vp]),
    "ethcnn_set_pass_pipeline": (_i, [_vp, _i]),
    "ethcnn_set_small_pass_launch": (_i, [_vp, _i]),
    "ethcnn_set_fc1_plan": (_i, [_vp, _i]),
    "ethcnn_get_fc1_plan": (_i, [_vp]),
    "ethcnn_check_fc1_plan": (_i, [_vp, _i, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "ethcnn_fast_plan_bound": (_i, [_vp, _sz, _i, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "ethcnn_measure_mfma_rate": (_i, [_vp, ctypes.c_double, ctypes.POINTER(ctypes.c_double)]),
    "ethcnn_ldp_step": (_i, [_vp, _vp, _i, _i, _pd, _i, _i, _vp, _fp]),
    "ethcnn_ldp_get_state": (_i, [_vp, _fp, _sz]),
    "ethcnn_ldp_step_begin": (_i, [_vp, _vp, _i, _i, _pd, _i, _i, _vp, _fp]),
    "ethcnn_rows_ready": (_i, [_vp, _i, _i]),
    "ethcnn_predict_luma_begin": (_i, [_vp, _vp, _i, _i, _i, _fp]),
    "ethcnn_predict_luma_end": (_i, [_vp]),
    "ethcnn_ldp_step_end": (_i, [_vp]),
    "ethcnn_host_alloc": (_i, [_vp, _sz, ctypes.POINTER(_vp)]),
    "ethcnn_host_free": (_i, [_vp, _vp]),
    "ethcnn_predict_luma": (_i, [_vp, _vp, _i, _i, _pd, _pd, _i, _i, _fp]),
    "ethcnn_predict_yuv_file": (_i, [_vp, _cp, _i, _i, _i, _cp, ctypes.POINTER(ctypes.c_int64)]),
    "ethcnn_predict_yuv_file_sharded": (_i, [_vp, ctypes.POINTER(ctypes.c_int), _i, _cp, _i, _i, _i, _cp, ctypes.POINTER(ctypes.c_int64)]),
    "ethcnn_shard_range": (_i, [ctypes.c_int64, _i, _i, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "ethcnn_get_startup_times": (_i, [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "ethcnn_predict_yuv_shard": (_i, [_vp, _cp, _i, _i, _i, _cp, ctypes.c_int64, ctypes.c_int64]),
    "ethcnn_predict_yuv_range": (_i, [_vp, _cp, _i, _i, _i, _cp, ctypes.c_int64, ctypes.c_int64]),
    "ethcnn_ckpt_read_blob": (_i, [_cp, _fp, _sz, ctypes.c_char_p, _sz]),
    "ethcnn_resi_vectors_device":(_i, [_vp, _vp, _i, _i, _pd, _vp]),
    "ethcnn_resi_vectors": (_i, [_vp, _vp, _i, _i, _pd, _fp]),
    "ethcnn_load_lstm_checkpoint": (_i, [_vp, _cp]),
    "ethcnn_load_lstm_blob": (_i, [_vp, _fp, _sz]),
    "ethcnn_load_lstm_synthetic": (_i, [_vp, ctypes.c_uint64, ctypes.c_double]),
    "ethcnn_get_lstm_blob": (_i, [_vp, _fp, _sz]),
    "ethcnn_lstm_model_name_for_qp": (_i, [_i, ctypes.c_char_p, _sz]),
    "ethcnn_lstm_step_device": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "ethcnn_ldp_predict_frame": (_i, [_vp, _vp, _i, _i, _pd, _i, _i, _vp, _fp, _fp]),
    "ethcnn_ckpt_read_lstm_blob": (_i, [_cp, _fp, _sz, ctypes.c_char_p, _sz]),
    "ethcnn_device_alloc": (_i, [_vp, _sz, ctypes.POINTER(_vp)]),
    "ethcnn_device_free": (_i, [_vp, _vp]),
    "ethcnn_memcpy_h2d": (_i, [_vp, _vp, _vp, _sz]),
    "ethcnn_memcpy_d2h": (_i, [_vp, _vp, _vp, _sz]),
    "ethcnn_synchronize": (_i, [_vp]),
    "ethcnn_device_name": (_i, [_vp, ctypes.c_char_p, _sz]),
    "ethcnn_set_profiling": (_i, [_vp, _i]),
    "ethcnn_get_stage_times": (_i, [_vp, ctypes.POINTER(StageTimes)]),
    "ethcnn_reset_stage_times": (_i, [_vp]),
    "ethcnn_debug_fetch": (_i, [_vp, _i, _fp, _sz]),
    "ethcnn_set_debug_capture": (_i, [_vp, _i]),
    "ethcnn_ckpt_read_index": (_i, [_cp, ctypes.POINTER(CkptEntry), _i, ctypes.POINTER(_i), ctypes.c_char_p, _sz]),
    "ethcnn_crc32c_masked": (ctypes.c_uint32, [_vp, _sz]),
    "ethcnn_host_thread_budget": (_i, [_i, _i]),
    "ethcnn_host_threads": (_i, [_vp]),
}

_lib = None


def load_library(path=None):
    """dlopen libethcnn.so and type every entry point.  Raises if it is not built."""
    global _lib
    if _lib is None or path is not None:
        p = path or LIB_PATH
        if not os.path.exists(p):
            raise OSError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no fallback implementation)" % p)
        lib = ctypes.CDLL(p)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        if path is not None:
            return lib
        _lib = lib
    return _lib


def model_name_for_qp(qp):
    buf = ctypes.create_string_buffer(64)
    rc = load_library().ethcnn_model_name_for_qp(int(qp), buf, 64)
    if rc:
        raise EthCnnError(rc, "model_name_for_qp")
    return buf.value.decode()


def parse_thresholds(thr_info_path):
    a, b = ctypes.c_float(), ctypes.c_float()
    rc = load_library().ethcnn_parse_thresholds(os.fsencode(thr_info_path), ctypes.byref(a), ctypes.byref(b))
    if rc:
        raise EthCnnError(rc, "cannot parse %s" % thr_info_path)
    return a.value, b.value


def read_ckpt_index(index_path):
    """[(name, dtype, shape, shard, offset, size, masked_crc32c)] of a TF-V2 .index file."""
    lib = load_library()
    ents = (CkptEntry * 256)()
    n = ctypes.c_int(0)
    err = ctypes.create_string_buffer(400)
    rc = lib.ethcnn_ckpt_read_index(index_path.encode(), ents, 256, ctypes.byref(n), err, 400)
    if rc:
        raise EthCnnError(rc, err.value.decode("utf-8", "replace"))
    # (names come from a file: a corrupted key need not be UTF-8 -- found by tests/test_ckpt.py's byte-flip fuzz)
    return [(e.name.decode("utf-8", "replace"), e.dtype, tuple(e.shape[i] for i in range(e.rank)), e.shard, e.offset, e.size, e.crc32c)
            for e in ents[: n.value]]


def lstm_model_name_for_qp(qp):
    buf = ctypes.create_string_buffer(64)
    rc = load_library().ethcnn_lstm_model_name_for_qp(int(qp), buf, 64)
    if rc:
        raise EthCnnError(rc, "lstm_model_name_for_qp")
    return buf.value.decode()


def read_ckpt_lstm_blob(prefix):
    """ETH-LSTM TF-V2 bundle -> float32[LSTM_BLOB_FLOATS] in checkpoint layout (crc32c-checked, host only)."""
    out = np.empty(LSTM_BLOB_FLOATS, dtype=np.float32)
    err = ctypes.create_string_buffer(400)
    rc = load_library().ethcnn_ckpt_read_lstm_blob(os.fsencode(prefix), out.ctypes.data_as(_fp), out.size, err, 400)
    if rc:
        raise EthCnnError(rc, err.value.decode("utf-8", "replace"))
    return out


def read_ckpt_blob(prefix):
    """TF-V2 bundle -> float32[BLOB_FLOATS] in checkpoint layout (crc32c-checked, host only)."""
    out = np.empty(BLOB_FLOATS, dtype=np.float32)
    err = ctypes.create_string_buffer(400)
    rc = load_library().ethcnn_ckpt_read_blob(os.fsencode(prefix), out.ctypes.data_as(_fp), out.size, err, 400)
    if rc:
        raise EthCnnError(rc, err.value.decode("utf-8", "replace"))
    return out


def crc32c_masked(data):
    b = bytes(data)
    return load_library().ethcnn_crc32c_masked(b, len(b))


def host_thread_budget(local_workers=1, usable_cpus=0):
    """staging-fill threads ONE predictor process starts when `local_workers` of them share the node (no device needed)"""
    return load_library().ethcnn_host_thread_budget(int(local_workers), int(usable_cpus))


def ctus_per_frame(width, height):
    return ((width + 63) // 64) * ((height + 63) // 64)


class DeviceBuffer(object):
    """HBM allocation owned by a context (ethcnn_device_alloc)."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = ctypes.c_void_p()
        ctx._chk(ctx.lib.ethcnn_device_alloc(ctx.h, self.nbytes, ctypes.byref(p)))
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.ctx._chk(self.ctx.lib.ethcnn_memcpy_h2d(self.ctx.h, self.ptr, arr.ctypes.data, arr.nbytes))

    def download(self, dtype, count):
        out = np.empty(int(count), dtype=dtype)
        assert out.nbytes <= self.nbytes
        self.ctx._chk(self.ctx.lib.ethcnn_memcpy_d2h(self.ctx.h, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            self.ctx.lib.ethcnn_device_free(self.ctx.h, self.ptr)
            self.ptr = None


class EthCnn(object):
    """One predictor context on one GPU (the reference's tf.Session + Saver + graph)."""

    def __init__(self, device=0, max_ctus_per_pass=0, host_threads=0):
        self.lib = load_library()
        opt = Options(device=int(device), max_ctus_per_pass=int(max_ctus_per_pass), host_threads=int(host_threads))
        h = ctypes.c_void_p()
        rc = self.lib.ethcnn_create(ctypes.byref(h), ctypes.byref(opt))
        if rc:
            raise EthCnnError(rc, self.lib.ethcnn_last_error(None).decode())
        self.h = h

    # -- plumbing
    def _chk(self, rc):
        if rc:
            raise EthCnnError(rc, self.lib.ethcnn_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.free_host_buffers()
            self.lib.ethcnn_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def device_name(self):
        buf = ctypes.create_string_buffer(128)
        self._chk(self.lib.ethcnn_device_name(self.h, buf, 128))
        return buf.value.decode()

    @property
    def host_threads(self):
        """staging-fill threads of this context (the node budget divided by the local worker count)"""
        n = self.lib.ethcnn_host_threads(self.h)
        if n < 0:
            self._chk(n)
        return n

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def synchronize(self):
        self._chk(self.lib.ethcnn_synchronize(self.h))

    # -- weights / thresholds
    def load_checkpoint(self, prefix):
        self._chk(self.lib.ethcnn_load_checkpoint(self.h, os.fsencode(prefix)))

    def load_blob(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        self._chk(self.lib.ethcnn_load_blob(self.h, blob.ctypes.data_as(_fp), blob.size))

    def load_synthetic(self, seed=1, head_gain=1.0):
        self._chk(self.lib.ethcnn_load_synthetic(self.h, int(seed), float(head_gain)))

    def get_blob(self):
        out = np.empty(BLOB_FLOATS, dtype=np.float32)
        self._chk(self.lib.ethcnn_get_blob(self.h, out.ctypes.data_as(_fp), out.size))
        return out

    def load_thresholds(self, thr_info_path):
        self._chk(self.lib.ethcnn_load_thresholds(self.h, os.fsencode(thr_info_path)))

    def set_thresholds(self, thr_l1_lower, thr_l2_lower):
        self._chk(self.lib.ethcnn_set_thresholds(self.h, thr_l1_lower, thr_l2_lower))

    def get_thresholds(self):
        a, b = ctypes.c_float(), ctypes.c_float()
        self._chk(self.lib.ethcnn_get_thresholds(self.h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    # -- prediction
    def predict_luma(self, luma, width, height, nframes, qp, pitch=None, frame_stride=None):
        """Host luma planes (uint8 buffer) -> float32 [nframes*nctu, 21]."""
        luma = np.ascontiguousarray(luma, dtype=np.uint8)
        pitch = width if pitch is None else pitch
        frame_stride = pitch * height if frame_stride is None else frame_stride
        need = (nframes - 1) * frame_stride + (height - 1) * pitch + width if nframes else 0
        if luma.size < need:
            raise ValueError("luma buffer too small: %d < %d" % (luma.size, need))
        out = np.empty((nframes * ctus_per_frame(width, height), NOUT), dtype=np.float32)
        self._chk(self.lib.ethcnn_predict_luma(self.h, luma.ctypes.data, width, height, pitch, frame_stride,
                                               nframes, int(qp), out.ctypes.data_as(_fp)))
        return out

    def predict_luma_begin(self, luma_pinned, width, height, qp, probs_out):
        """streamed input (ethcnn_predict_luma_begin): queue ONE picture's pass on a host_buffer() the caller is still filling; report
        CTU rows with rows_ready (any thread), finish with predict_luma_end.  probs_out: float32 array of nctu * 21 (kept alive by
        the caller until predict_luma_end returns)"""
        n = ctus_per_frame(width, height)
        assert luma_pinned.dtype == np.uint8 and luma_pinned.flags["C_CONTIGUOUS"] and luma_pinned.size >= width * height
        assert probs_out.dtype == np.float32 and probs_out.size == n * NOUT and probs_out.flags["C_CONTIGUOUS"]
        self._chk(self.lib.ethcnn_predict_luma_begin(self.h, luma_pinned.ctypes.data, width, height, int(qp), probs_out.ctypes.data_as(_fp)))

    def predict_luma_end(self):
        self._chk(self.lib.ethcnn_predict_luma_end(self.h))

    def predict_luma_device(self, d_luma, width, height, nframes, qp, d_probs, pitch=None, frame_stride=None):
        """Both pointers already in HBM (ints or DeviceBuffer); asynchronous."""
        pitch = width if pitch is None else pitch
        frame_stride = pitch * height if frame_stride is None else frame_stride
        src = d_luma.ptr if isinstance(d_luma, DeviceBuffer) else int(d_luma)
        dst = d_probs.ptr if isinstance(d_probs, DeviceBuffer) else int(d_probs)
        self._chk(self.lib.ethcnn_predict_luma_device(self.h, src, width, height, pitch, frame_stride, nframes,
                                                      int(qp), dst))

    def predict_ctus(self, ctus, qp):
        """[n,64,64] uint8 CTUs -> [n,21]; gates per <=1024-CTU sub-batch exactly like
        get_y_conv_on_large_data (video_to_cu_depth.py:61-73): a 64-wide, 64n-tall 'frame'."""
        ctus = np.ascontiguousarray(ctus, dtype=np.uint8).reshape(-1, 64, 64)
        n = ctus.shape[0]
        if n == 0:
            return np.zeros((0, NOUT), dtype=np.float32)
        return self.predict_luma(ctus.reshape(-1), 64, 64 * n, 1, qp)

    def predict_yuv_file(self, yuv_path, width, height, qp, out_path):
        nf = ctypes.c_int64(0)
        self._chk(self.lib.ethcnn_predict_yuv_file(self.h, os.fsencode(yuv_path), width, height, int(qp),
                                                   os.fsencode(out_path), ctypes.byref(nf)))
        return nf.value

    def predict_yuv_file_sharded(self, devices, yuv_path, width, height, qp, out_path):
        """the whole file over `devices` (HIP ordinals; devices[0] = this context's) from this process: a worker thread per entry
        (ethcnn_predict_yuv_file_sharded; ctypes releases the GIL for the call)"""
        nf = ctypes.c_int64(0)
        arr = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        self._chk(self.lib.ethcnn_predict_yuv_file_sharded(self.h, arr, len(devices), os.fsencode(yuv_path), width, height, int(qp),
                                                           os.fsencode(out_path), ctypes.byref(nf)))
        return nf.value

    def startup_times(self):
        """(ms of ethcnn_create's first HIP call = runtime initialisation, ms of the whole ethcnn_create)"""
        a, b = ctypes.c_double(), ctypes.c_double()
        self._chk(self.lib.ethcnn_get_startup_times(self.h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def predict_yuv_range(self, yuv_path, width, height, qp, out_path, frame_begin, frame_end):
        """frames [frame_begin, frame_end) -> an out_path holding exactly those (get_prob's n_frames_start / n_frames_end)"""
        self._chk(self.lib.ethcnn_predict_yuv_range(self.h, os.fsencode(yuv_path), width, height, int(qp),
                                                    os.fsencode(out_path), int(frame_begin), int(frame_end)))
        return int(frame_end) - int(frame_begin)

    def predict_yuv_shard(self, yuv_path, width, height, qp, out_path, frame_begin, frame_end):
        self._chk(self.lib.ethcnn_predict_yuv_shard(self.h, os.fsencode(yuv_path), width, height, int(qp),
                                                    os.fsencode(out_path), int(frame_begin), int(frame_end)))

    def resi_vectors(self, luma, width, height, pitch=None):
        luma = np.ascontiguousarray(luma, dtype=np.uint8)
        pitch = width if pitch is None else pitch
        need = (height - 1) * pitch + width if width > 0 and height > 0 else 0
        if luma.size < need:
            raise ValueError("luma buffer too small: %d < %d" % (luma.size, need))
        out = np.empty((ctus_per_frame(width, height), NVEC), dtype=np.float32)
        self._chk(self.lib.ethcnn_resi_vectors(self.h, luma.ctypes.data, width, height, pitch, out.ctypes.data_as(_fp)))
        return out

    def resi_vectors_device(self, d_luma, width, height, d_vec, pitch=None):
        """asynchronous: LDP front-end on device buffers (ethcnn_resi_vectors_device)"""
        pitch = width if pitch is None else pitch
        self._chk(self.lib.ethcnn_resi_vectors_device(self.h, d_luma.ptr, width, height, pitch, d_vec.ptr))

    # -- config #5 back-end: ETH-LSTM one step (resi_to_cu_depth_LDP.py:108-129)
    def load_lstm_checkpoint(self, prefix):
        self._chk(self.lib.ethcnn_load_lstm_checkpoint(self.h, os.fsencode(prefix)))

    def load_lstm_blob(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        self._chk(self.lib.ethcnn_load_lstm_blob(self.h, blob.ctypes.data_as(_fp), blob.size))

    def load_lstm_synthetic(self, seed=1, head_gain=1.0):
        self._chk(self.lib.ethcnn_load_lstm_synthetic(self.h, int(seed), float(head_gain)))

    def get_lstm_blob(self):
        out = np.empty(LSTM_BLOB_FLOATS, dtype=np.float32)
        self._chk(self.lib.ethcnn_get_lstm_blob(self.h, out.ctypes.data_as(_fp), out.size))
        return out

    def lstm_step(self, vec, state_in, qp, i_frame):
        """vec [n,448], state_in [n,2,448] or None -> (probs [n,21] gated, state_out [n,2,448]); the
        vectors go through HBM buffers and the *_device entry point."""
        vec = np.ascontiguousarray(vec, dtype=np.float32)
        n = vec.shape[0]
        dv, dso, dp = self.alloc(vec.nbytes), self.alloc(n * 2 * NVEC * 4), self.alloc(n * NOUT * 4)
        dv.upload(vec)
        dsi = None
        if state_in is not None:
            state_in = np.ascontiguousarray(state_in, dtype=np.float32).reshape(n, 2, NVEC)
            dsi = self.alloc(state_in.nbytes)
            dsi.upload(state_in)
        self._chk(self.lib.ethcnn_lstm_step_device(self.h, dv.ptr, dsi.ptr if dsi else None, n, int(qp), int(i_frame),
                                                   dso.ptr, dp.ptr))
        probs = dp.download(np.float32, n * NOUT).reshape(n, NOUT)
        state = dso.download(np.float32, n * 2 * NVEC).reshape(n, 2, NVEC)
        for b in (dv, dso, dp, dsi):
            if b is not None:
                b.free()
        return probs, state

    def ldp_predict_frame(self, luma, width, height, qp, i_frame, state_in=None, pitch=None):
        """one frame of resi.yuv luma -> (probs [nctu,21], state_out [nctu,2,448])"""
        luma = np.ascontiguousarray(luma, dtype=np.uint8)
        pitch = width if pitch is None else pitch
        need = (height - 1) * pitch + width if width > 0 and height > 0 else 0
        if luma.size < need:
            raise ValueError("luma buffer too small: %d < %d" % (luma.size, need))
        n = ctus_per_frame(width, height)
        probs = np.empty((n, NOUT), dtype=np.float32)
        state = np.empty((n, 2, NVEC), dtype=np.float32)
        sin = None
        if state_in is not None:
            sin = np.ascontiguousarray(state_in, dtype=np.float32).reshape(n, 2, NVEC)
        self._chk(self.lib.ethcnn_ldp_predict_frame(self.h, luma.ctypes.data, width, height, pitch, int(qp), int(i_frame),
                                                    sin.ctypes.data if sin is not None else None,
                                                    state.ctypes.data_as(_fp), probs.ctypes.data_as(_fp)))
        return probs, state

    def ldp_step(self, luma, width, height, qp, i_frame, state_in=None, pitch=None, probs_out=None):
        """one frame with the recurrent state resident in HBM between calls (ethcnn_ldp_step): state_in None = the
        previous call's state (zeros when i_frame <= 1) -> probs [nctu, 21]"""
        luma = np.ascontiguousarray(luma, dtype=np.uint8)
        pitch = width if pitch is None else pitch
        need = (height - 1) * pitch + width if width > 0 and height > 0 else 0
        if luma.size < need:
            raise ValueError("luma buffer too small: %d < %d" % (luma.size, need))
        n = ctus_per_frame(width, height)
        probs = np.empty((n, NOUT), dtype=np.float32) if probs_out is None else probs_out
        assert probs.dtype == np.float32 and probs.size == n * NOUT and probs.flags["C_CONTIGUOUS"]
        sin = None
        if state_in is not None:
            sin = np.ascontiguousarray(state_in, dtype=np.float32).reshape(n, 2, NVEC)
        self._chk(self.lib.ethcnn_ldp_step(self.h, luma.ctypes.data, width, height, pitch, int(qp), int(i_frame),
                                           sin.ctypes.data if sin is not None else None, probs.ctypes.data_as(_fp)))
        return probs.reshape(n, NOUT)

    def ldp_step_begin(self, luma_pinned, width, height, qp, i_frame, probs_out, state_in=None, pitch=None):
        """streamed input (ethcnn_ldp_step_begin): queue the frame's kernels on a host_buffer() the caller is still filling; report
        rows with rows_ready (any thread), finish with ldp_step_end.  probs_out: float32 array of nctu * 21 (kept alive by the
        caller until ldp_step_end returns)"""
        pitch = width if pitch is None else pitch
        n = ctus_per_frame(width, height)
        assert luma_pinned.dtype == np.uint8 and luma_pinned.flags["C_CONTIGUOUS"] and luma_pinned.size >= (height - 1) * pitch + width
        assert probs_out.dtype == np.float32 and probs_out.size == n * NOUT and probs_out.flags["C_CONTIGUOUS"]
        sin = None
        if state_in is not None:
            sin = np.ascontiguousarray(state_in, dtype=np.float32).reshape(n, 2, NVEC)
        self._chk(self.lib.ethcnn_ldp_step_begin(self.h, luma_pinned.ctypes.data, width, height, pitch, int(qp), int(i_frame),
                                                 sin.ctypes.data if sin is not None else None, probs_out.ctypes.data_as(_fp)))

    def rows_ready(self, ctu_row_begin, ctu_row_end):
        if self.lib.ethcnn_rows_ready(self.h, int(ctu_row_begin), int(ctu_row_end)) != 0:
            raise ValueError("rows_ready(%d, %d)" % (ctu_row_begin, ctu_row_end))

    def ldp_step_end(self):
        self._chk(self.lib.ethcnn_ldp_step_end(self.h))

    def ldp_get_state(self, width, height):
        n = ctus_per_frame(width, height)
        state = np.empty((n, 2, NVEC), dtype=np.float32)
        self._chk(self.lib.ethcnn_ldp_get_state(self.h, state.ctypes.data_as(_fp), state.size))
        return state

    def host_buffer(self, nbytes):
        """pinned host memory as a uint8 numpy array (freed with the context, or by free_host_buffer)"""
        p = _vp()
        self._chk(self.lib.ethcnn_host_alloc(self.h, int(nbytes), ctypes.byref(p)))
        arr = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(max(1, int(nbytes)),))
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p.value)
        return arr[:nbytes]

    def free_host_buffers(self):
        for ptr in getattr(self, "_pinned", []):
            self.lib.ethcnn_host_free(self.h, ptr)
        self._pinned = []

    # -- measurement / introspection
    def set_profiling(self, level=2):
        """0 off, 1 dominant kernel (FC1) only, 2 every stage."""
        self._chk(self.lib.ethcnn_set_profiling(self.h, int(level)))

    def set_pass_pipeline(self, on=True):
        """CTU-load stage of pass i+1 beside FC1 of pass i (default on); off = one stream, stage timings do not overlap"""
        self._chk(self.lib.ethcnn_set_pass_pipeline(self.h, 1 if on else 0))

    def measure_mfma_rate(self, seconds=0.05):
        """TFLOP/s of pure exact-fp32 MFMAs this GPU sustains (box calibration for reading roofline fractions)"""
        v = ctypes.c_double(0.0)
        self._chk(self.lib.ethcnn_measure_mfma_rate(self.h, float(seconds), ctypes.byref(v)))
        return v.value

    def set_fc1_plan(self, plan=3):
        """arithmetic plan of big passes: 0 = exact fp32 (default, bit-identical to the oracle); 2 = FC1 as two-way fp16 splits on the
        16-bit matrix pipe; 3 = trunk, FC1 and heads that way (as accurate against float64, NOT bit-identical; include/ethcnn.h)."""
        self._chk(self.lib.ethcnn_set_fc1_plan(self.h, int(plan)))

    def fc1_plan(self):
        return int(self.lib.ethcnn_get_fc1_plan(self.h))

    def check_fc1_plan(self, plan):
        """the load-time accuracy guard of plans 2 / 3 for the loaded weights (ethcnn_check_fc1_plan) ->
        {"accepted", "apriori_bound", "measured" (None: the a-priori bound sufficed), "message"}; never raises for a refusal"""
        b, m = ctypes.c_double(0.0), ctypes.c_double(-1.0)
        rc = self.lib.ethcnn_check_fc1_plan(self.h, int(plan), ctypes.byref(b), ctypes.byref(m))
        if rc not in (0, ERR_PLAN_REFUSED):
            self._chk(rc)
        return {"accepted": rc == 0, "apriori_bound": b.value, "measured": None if m.value < 0 else m.value,
                "message": "" if rc == 0 else self.lib.ethcnn_last_error(self.h).decode()}

    def set_small_pass_launch(self, on=True):
        """one picture (<= 2304 CTUs, 16-byte aligned rows) as ONE launch (default on); off = five launches.  Same results."""
        self._chk(self.lib.ethcnn_set_small_pass_launch(self.h, 1 if on else 0))

    def reset_stage_times(self):
        self._chk(self.lib.ethcnn_reset_stage_times(self.h))

    def stage_times(self):
        st = StageTimes()
        self._chk(self.lib.ethcnn_get_stage_times(self.h, ctypes.byref(st)))
        return {"ms": dict(zip(STAGES, list(st.ms))), "launches": dict(zip(STAGES, list(st.launches))), "ctus": st.ctus,
                "timed": dict(zip(STAGES, list(st.timed))), "timed_ctus": dict(zip(STAGES, list(st.timed_ctus))),
                "timing_errors": st.timing_errors}

    def set_debug_capture(self, on=True):
        """store FC2 outputs, logits and ungated probabilities of the following passes (debug_fetch)"""
        self._chk(self.lib.ethcnn_set_debug_capture(self.h, 1 if on else 0))

    def debug_fetch(self, which, n):
        out = np.empty((n, _DBG_WIDTH[which]), dtype=np.float32)
        self._chk(self.lib.ethcnn_debug_fetch(self.h, which, out.ctypes.data_as(_fp), out.size))
        return out

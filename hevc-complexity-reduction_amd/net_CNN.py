"""Host mirror of the reference's network module
(/root/reference/HM-16.5_Test_AI/bin/net_CNN.py): same names, same argument meaning.

The reference builds a TensorFlow graph here; this module only holds the constants, the
Thr_info.txt reader and a `net()` that runs the whole graph for one fed batch on the GPU
through libethcnn.so.  No arithmetic happens in Python.
"""
import numpy as np

from . import ethcnn as _e

# net_CNN.py:8-36
IMAGE_SIZE = 64
NUM_CHANNELS = 1
NUM_EXT_FEATURES = 1
NUM_LABEL_BYTES = 16
NUM_CONVLAYER1_FILTERS = 16
NUM_CONVLAYER2_FILTERS = 24
NUM_CONVLAYER3_FILTERS = 32
NUM_CONV2_FLAT_S_FILTERS = 8 * 8 * NUM_CONVLAYER2_FILTERS
NUM_CONV2_FLAT_M_FILTERS = 4 * 4 * NUM_CONVLAYER2_FILTERS
NUM_CONV2_FLAT_L_FILTERS = 2 * 2 * NUM_CONVLAYER2_FILTERS
NUM_CONV3_FLAT_S_FILTERS = 4 * 4 * NUM_CONVLAYER3_FILTERS
NUM_CONV3_FLAT_M_FILTERS = 2 * 2 * NUM_CONVLAYER3_FILTERS
NUM_CONV3_FLAT_L_FILTERS = 1 * 1 * NUM_CONVLAYER3_FILTERS
NUM_CONVLAYER_FLAT_FILTERS = (NUM_CONV2_FLAT_S_FILTERS + NUM_CONV2_FLAT_M_FILTERS + NUM_CONV2_FLAT_L_FILTERS +
                              NUM_CONV3_FLAT_S_FILTERS + NUM_CONV3_FLAT_M_FILTERS + NUM_CONV3_FLAT_L_FILTERS)
NUM_DENLAYER1_FEATURES_64, NUM_DENLAYER2_FEATURES_64 = 64, 48
NUM_DENLAYER1_FEATURES_32, NUM_DENLAYER2_FEATURES_32 = 128, 96
NUM_DENLAYER1_FEATURES_16, NUM_DENLAYER2_FEATURES_16 = 256, 192
assert NUM_CONVLAYER_FLAT_FILTERS == _e.NFEAT


def get_thresholds(thr_file):
    """net_CNN.py:38-45 -- tokens [1] and [3] of the first line split on single spaces.
    Parsed by the library (ethcnn_load_thresholds) so the CLI and this mirror agree."""
    return _e.parse_thresholds(thr_file)


def net(ctx, x, qp):
    """One fed batch (net_CNN.py:103-195 for what sess.run fetches at
    video_to_cu_depth.py:71): x [n,64,64(,1)] pixel values 0..255, qp the sequence QP.
    Returns (y_conv_flat_64 [n,1], y_conv_flat_32 [n,4], y_conv_flat_16 [n,16]) with the
    batch-level gates (:175,187) applied over THIS batch, so n must not exceed the
    reference's sub-batch of 1024."""
    x = np.asarray(x)
    n = x.shape[0]
    if n > _e.SUB_BATCH:
        raise ValueError("net(): one fed batch holds at most %d CTUs (video_to_cu_depth.py:64)" % _e.SUB_BATCH)
    if x.dtype != np.uint8:
        xi = np.rint(x).astype(np.int64)
        if np.any(xi != x) or xi.min(initial=0) < 0 or xi.max(initial=0) > 255:
            raise ValueError("net(): x must hold 8-bit pixel values")
        x = xi.astype(np.uint8)
    y = ctx.predict_ctus(x.reshape(n, 64, 64), qp)
    return y[:, 0:1], y[:, 1:5], y[:, 5:21]

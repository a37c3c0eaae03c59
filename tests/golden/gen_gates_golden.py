"""Generates tests/golden/gates_golden.npz: a fixture for the batch gates (net_CNN.py:175,187), which no reference-derived
vector covered (the saved .meta graphs have no gates).  The expected outputs come from tests/gates_ref.py -- a 30-line numpy
evaluation of the two tf.cond lines over the fed sub-batches, independent of the oracle and of the HIP code -- applied to the
UNGATED probabilities of a seeded sequence.  Threshold cases include values EXACTLY at a sub-batch maximum (strict `>` must
leave that sub-batch closed), one ulp below it (open), a closed L1 gate with `0 > thr2` true (y32 zeroed but y16 kept) and
false, and everything open.  The file holds the geometry, the generator seed of the pixels, the ungated probabilities, the
threshold pairs and the expected gated outputs; consumers: tests/test_gates_golden.py (oracle on CPU, HIP path with -m gpu).
Geometry: 2 frames of 64 x (64 * 1100) pixels = 1100 CTUs per frame = sub-batches of 1024 + 76 (the ragged tail)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, HERE)
import ctu_gen  # noqa: E402
import ethcnn_np as oracle  # noqa: E402  (ungated probabilities only: thresholds -1 -> every gate open)
import gates_ref  # noqa: E402

SEED, WSEED, GAIN, QP, NCTU, NFRAMES = 4242, 5, 2.0, 32, 1100, 2


def luma_strip():
    """the four sub-batches (1024 + 76 per frame) draw from different CTU classes of ctu_gen, so their maxima differ"""
    pool = ctu_gen.make_ctus(SEED, 8 * 1024)
    allowed = [(0, 1, 2, 3, 4, 5, 6, 7), (0, 1, 7), (1, 2, 7), (0, 7)]
    bounds = [(0, 1024), (1024, 1100), (1100, 2124), (2124, 2200)]
    out = np.empty((NCTU * NFRAMES, 64, 64), dtype=np.uint8)
    for k, (a, b) in enumerate(bounds):
        idx = [i for i in range(pool.shape[0]) if i % 8 in allowed[k]]
        out[a:b] = pool[[idx[(j * 7 + k) % len(idx)] for j in range(b - a)]]
    return out.reshape(NFRAMES, NCTU * 64, 64)


def main():
    blob = oracle.synth_blob(WSEED, GAIN)
    luma = luma_strip()
    raw = oracle.predict_frames(blob, luma, 64, 64 * NCTU, NFRAMES, QP, -1.0, -1.0, mode=0)
    assert raw.shape == (NFRAMES * NCTU, 21)
    sub = [(0, 1024), (1024, 1100), (1100, 2124), (2124, 2200)]
    m64 = [float(raw[a:b, 0].max()) for a, b in sub]
    m32 = [float(raw[a:b, 1:5].max()) for a, b in sub]
    dn = lambda v: float(np.nextafter(np.float32(v), np.float32(-1)))
    cases = [(0.5, 0.5),
             (m64[0], 0.5), (dn(m64[0]), 0.5),      # L1 exactly at / one ulp below the maximum of sub-batch 0 of frame 0
             (m64[3], 0.5),                         # ... of the ragged tail of frame 1
             (m64[1], 0.5), (dn(m64[2]), 0.45),
             (max(m64), 0.5),                       # at the global maximum: every sub-batch closed
             (0.0, m32[2]), (0.0, dn(m32[2])),      # L2 exactly at / below the y32 maximum of sub-batch 0 of frame 1
             (2.0, -0.5),                           # L1 closed everywhere, 0 > thr2 is TRUE: y32 zeros, y16 KEPT
             (2.0, 0.0),                            # L1 closed, 0 > 0 false: y16 zeros too
             (-1.0, -1.0)]                          # everything open
    thr = np.array(cases, dtype=np.float32)
    want = np.stack([gates_ref.gate_frames(raw, NCTU, t1, t2) for t1, t2 in thr])
    closed = [[int(not want[i, a:b, 1:5].any()) + 2 * int(not want[i, a:b, 5:].any()) for a, b in sub] for i in range(len(cases))]
    print("sub-batch states (bit 0: y32 zeroed, bit 1: y16 zeroed):")
    for c, s in zip(cases, closed):
        print("  thr = (%.9g, %.9g): %s" % (c[0], c[1], s))
    import zlib
    # the expected outputs are `raw` with whole sub-batch blocks zeroed: the file keeps the block states + a crc per case
    np.savez_compressed(os.path.join(HERE, "gates_golden.npz"), params=np.array([SEED, WSEED, GAIN, QP, NCTU, NFRAMES], dtype=np.float64),
                        luma_crc=np.int64(ctu_gen.crc(luma)), raw=raw, thr=thr, states=np.array(closed, dtype=np.int8),
                        want_crc=np.array([zlib.crc32(w.tobytes()) for w in want], dtype=np.int64))


if __name__ == "__main__":
    main()

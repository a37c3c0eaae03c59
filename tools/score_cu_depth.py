#!/usr/bin/env python
"""Scores a predicted cu_depth.dat against ground-truth CU depths.

Ground truth: the reference's label files `Info_*_CUDepth.dat` (one byte per 16x16 block = CU depth 0..3,
raster order, frame after frame; written by HM-16.5_Extract_Data, TEncSlice.cpp:880-888,905-921; examples in
/root/reference/AI_Info/).  Prediction: cu_depth.dat = float32 [frames][ctus][21] as written by
video_to_cu_depth.py / libethcnn.

The score is the hierarchical classification of the reference's training script
(ETH-CNN_Training_AI/train_CNN_CTU64.py:103-137 get_class_matrices, :65-96 is_sep_*; truth thresholds
input_data.DEFAULT_THR_LIST = [0.5, 1.5, 2.5]): per CTU "split 64?" (mean depth of the 16 blocks > 0.5 vs
p64 > thr), per 32x32 of a truly split CTU "split 32?" (mean depth of its 4 blocks > 1.5 vs p32 > thr), per
16x16 of a truly split 32x32 "split 16?" (depth > 2.5 vs p16 > thr).  Vectorized numpy; host-side tooling,
not part of the hot path.

    score_cu_depth.py <Info_..._CUDepth.dat> <cu_depth.dat> <width> <height> [thr64 thr32 thr16]
"""
import sys

import numpy as np

TRUTH_THR = (0.5, 1.5, 2.5)
IDX32 = np.array([[0, 1, 4, 5], [2, 3, 6, 7], [8, 9, 12, 13], [10, 11, 14, 15]])  # 16x16 blocks of each 32x32, raster


def read_labels(path, width, height):
    """-> uint8 [frames, height/16, width/16]"""
    assert width % 16 == 0 and height % 16 == 0, "label files exist for sizes that are multiples of 16"
    per = (width // 16) * (height // 16)
    raw = np.fromfile(path, dtype=np.uint8)
    assert raw.size % per == 0, "%s: %d bytes is not a multiple of %d blocks per frame" % (path, raw.size, per)
    assert raw.max(initial=0) <= 3, "CU depths are 0..3"
    return raw.reshape(-1, height // 16, width // 16)


def labels_per_ctu(labels):
    """[frames, H16, W16] -> [frames * ctus, 16]: the 4x4 depth map of every whole CTU in raster order
    (the layout of the 16 label bytes of a training sample, Extract_Data/extract_data_AI.py:94-111)"""
    f, h16, w16 = labels.shape
    assert h16 % 4 == 0 and w16 % 4 == 0, "frame size must be a multiple of 64 (whole CTUs)"
    return labels.reshape(f, h16 // 4, 4, w16 // 4, 4).transpose(0, 1, 3, 2, 4).reshape(-1, 16)


def class_matrices(depth16, probs21, thr=(0.5, 0.5, 0.5)):
    """depth16 [n,16] ints 0..3, probs21 [n,21] -> three 2x2 matrices m[truth][predicted] (64, 32, 16)"""
    depth16 = np.asarray(depth16, dtype=np.float64)
    p = np.asarray(probs21, dtype=np.float64)
    assert depth16.shape[0] == p.shape[0] and depth16.shape[1] == 16 and p.shape[1] == 21
    out = []
    t64 = depth16.mean(axis=1) > TRUTH_THR[0]
    q64 = p[:, 0] > thr[0]
    out.append(_matrix(t64, q64))
    d32 = depth16[:, IDX32]                      # [n, 4 (32x32), 4 (16x16)]
    t32 = d32.mean(axis=2) > TRUTH_THR[1]        # [n,4]
    q32 = p[:, 1:5] > thr[1]
    m32 = np.broadcast_to(t64[:, None], t32.shape)
    out.append(_matrix(t32[m32], q32[m32]))
    t16 = d32 > TRUTH_THR[2]                     # [n,4,4]
    q16 = p[:, 5:][:, IDX32] > thr[2]
    m16 = np.broadcast_to((t64[:, None] & t32)[:, :, None], t16.shape)
    out.append(_matrix(t16[m16], q16[m16]))
    return out


def _matrix(truth, pred):
    truth, pred = np.asarray(truth, dtype=bool), np.asarray(pred, dtype=bool)
    return [[int((~truth & ~pred).sum()), int((~truth & pred).sum())], [int((truth & ~pred).sum()), int((truth & pred).sum())]]


def accuracy(m):
    tot = m[0][0] + m[0][1] + m[1][0] + m[1][1]
    return (m[0][0] + m[1][1]) / tot if tot else float("nan")


def main(argv):
    if len(argv) not in (5, 8):
        sys.stderr.write(__doc__)
        return 2
    w, h = int(argv[3]), int(argv[4])
    thr = tuple(float(x) for x in argv[5:8]) if len(argv) == 8 else (0.5, 0.5, 0.5)
    depth = labels_per_ctu(read_labels(argv[1], w, h))
    probs = np.fromfile(argv[2], dtype="<f4").reshape(-1, 21)
    n = min(depth.shape[0], probs.shape[0])
    if depth.shape[0] != probs.shape[0]:
        sys.stderr.write("note: %d labelled CTUs, %d predicted CTUs: scoring the first %d\n" % (depth.shape[0], probs.shape[0], n))
    for name, m in zip(("64x64", "32x32", "16x16"), class_matrices(depth[:n], probs[:n], thr)):
        print("%s  [[n00 n01] [n10 n11]] = %s  accuracy %.4f" % (name, m, accuracy(m)))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))

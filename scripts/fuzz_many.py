"""One-off wide fuzz (GPU box): N random geometries / strides / thresholds / QPs, AI + resi + LDP, bit-exact vs the oracle."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import ethcnn_np as oracle, ethcnn_lstm_np as ol
pkg = importlib.import_module("hevc-complexity-reduction_amd")
N = int(os.environ.get("CASES", "300"))
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
ctx = pkg.EthCnn(0)
bad = 0
t0 = time.time()
for case in range(N):
    if case % 25 == 0:
        blob = oracle.synth_blob(int(rng.integers(1, 1 << 30)), float(rng.choice([1.0, 4.0, 8.0, 16.0])))
        lblob = ol.synth_lstm_blob(int(rng.integers(1, 1 << 30)), float(rng.choice([1.0, 3.0, 6.0])))
        ctx.load_blob(blob); ctx.load_lstm_blob(lblob)
    big = case % 10 == 0
    w = int(rng.integers(1, 2200 if big else 600)); h = int(rng.integers(1, 1300 if big else 400))
    pitch = w + int(rng.choice([0, 0, 3, 16, 100]))
    frames = int(rng.integers(1, 3 if big else 6))
    stride = pitch * h + int(rng.choice([0, 5, pitch * (h // 2)]))
    qp = int(rng.integers(10, 52))
    thr = [float(x) for x in rng.choice([-1.0, 0.0, 0.2, 0.4, 0.5, 0.6, 0.8, 0.99, 1.5], size=2)]
    luma = rng.integers(0, 256, size=stride * frames + 8, dtype=np.uint8)
    mode = case % 4
    if mode == 1: luma[:] = luma // 32 + 110
    if mode == 2: luma[: luma.size // 2] = 0
    ctx.set_thresholds(*thr)
    got = ctx.predict_luma(luma, w, h, frames, qp, pitch=pitch, frame_stride=stride)
    want = oracle.predict_frames(blob, luma, w, h, frames, qp, thr[0], thr[1], mode=0, pitch=pitch, frame_stride=stride)
    ok = np.array_equal(got.view(np.uint32), want.view(np.uint32))
    if case % 3 == 0:  # LDP chain of 2 frames on the first plane
        plane = np.ascontiguousarray(luma[: pitch * h].reshape(h, pitch)[:, :w])
        gs = os_ = None
        for i in (1, 2):
            gp, gs = ctx.ldp_predict_frame(plane, w, h, qp, i, gs)
            op, os_ = ol.lstm_step(lblob, oracle.resi_vectors(blob, plane, w, h), os_, qp, i, thr[0], thr[1], mode=0)
            ok = ok and np.array_equal(gp.view(np.uint32), op.view(np.uint32)) and np.array_equal(gs.view(np.uint32), os_.view(np.uint32))
    if not ok:
        bad += 1
        print("MISMATCH case", case, w, h, pitch, stride, frames, qp, thr)
print("fuzz: %d cases, %d mismatches, %.0f s" % (N, bad, time.time() - t0))
sys.exit(1 if bad else 0)

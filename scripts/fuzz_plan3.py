"""One-off wide fuzz of the opt-in plans (GPU box): N random geometries / pitches / strides / QPs big enough for the multi-launch path,
plan 2 and plan 3 against the oracle at the north star's 1e-4 (thresholds that leave no knife edges: every gate open, or closed)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import ethcnn_np as oracle
pkg = importlib.import_module("hevc-complexity-reduction_amd")
N = int(os.environ.get("CASES", "120"))
rng = np.random.default_rng(int(os.environ.get("SEED", "5")))
ctx = pkg.EthCnn(0)
bad, worst = 0, 0.0
t0 = time.time()
for case in range(N):
    if case % 10 == 0:
        blob = oracle.synth_blob(int(rng.integers(1, 1 << 30)), float(rng.choice([1.0, 4.0, 8.0])))
        ctx.load_blob(blob)
    aligned = case % 3 != 0
    w = int(rng.integers(300, 2600)); h = int(rng.integers(200, 1500))
    if aligned: w = (w + 15) // 16 * 16
    nctu = ((w + 63) // 64) * ((h + 63) // 64)
    frames = max(1, -(-2400 // nctu)) + int(rng.integers(0, 3))  # > 2304 CTUs: the multi-launch path, where the plans apply
    pitch = w + (int(rng.choice([0, 16, 64])) if aligned else int(rng.choice([0, 3, 100])))
    stride = pitch * h + (int(rng.choice([0, 16 * pitch])) if aligned else int(rng.choice([0, 5])))
    qp = int(rng.integers(10, 52))
    thr = [-1.0, -1.0] if case % 2 == 0 else [float(rng.choice([-1.0, 1.5])), float(rng.choice([-1.0, 1.5]))]
    luma = rng.integers(0, 256, size=stride * frames + 16, dtype=np.uint8)
    mode = case % 4
    if mode == 1: luma[:] = luma // 32 + 110
    if mode == 2: luma[: luma.size // 2] = 0
    if aligned and case % 2: luma = np.concatenate([np.zeros(16, np.uint8), luma])[16:]  # (a copy: keeps the base pointer's alignment arbitrary)
    ctx.set_thresholds(*thr)
    want = oracle.predict_frames(blob, luma, w, h, frames, qp, thr[0], thr[1], mode=0, pitch=pitch, frame_stride=stride)
    for plan in (2, 3):
        ctx.set_fc1_plan(plan)
        got = ctx.predict_luma(luma, w, h, frames, qp, pitch=pitch, frame_stride=stride)
        d = float(np.abs(got - want).max())
        worst = max(worst, d)
        if not (np.array_equal(got == 0, want == 0) and d <= 1e-4):
            bad += 1
            print("MISMATCH case", case, "plan", plan, w, h, pitch, stride, frames, qp, thr, "max|d|", d)
    ctx.set_fc1_plan(0)
    got = ctx.predict_luma(luma, w, h, frames, qp, pitch=pitch, frame_stride=stride)
    if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
        bad += 1
        print("MISMATCH case", case, "plan 0 (bit-exact)", w, h, pitch, stride, frames, qp, thr)
print("fuzz plans 2 / 3: %d cases, %d mismatches, worst max|d| %.3g, %.0f s" % (N, bad, worst, time.time() - t0))
sys.exit(1 if bad else 0)

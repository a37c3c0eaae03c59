"""Race / determinism soak: many repeated passes over the same inputs (AI path, LDP path, two contexts in
two threads) must give bit-identical outputs every time, and equal the oracle.  GPU box."""
import importlib, os, sys, threading, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import ethcnn_np as oracle, ethcnn_lstm_np as ol
pkg = importlib.import_module("hevc-complexity-reduction_amd")
REPS = int(os.environ.get("REPS", "150"))
bad = []

def ai_case(w, h, frames, qp, seed, reps):
    rng = np.random.default_rng(seed)
    blob = oracle.synth_blob(seed, 8.0)
    luma = rng.integers(0, 256, size=(frames, h, w), dtype=np.uint8)
    want = oracle.predict_frames(blob, luma, w, h, frames, qp, 0.5, 0.5, mode=0)
    c = pkg.EthCnn(0)
    c.load_blob(blob)
    d_in, d_out = c.alloc(luma.nbytes), c.alloc(want.nbytes)
    d_in.upload(luma)
    n = want.size
    for r in range(reps):
        c.predict_luma_device(d_in, w, h, frames, qp, d_out)
        if r % 3 == 0:
            got = d_out.download(np.float32, n)
            if not np.array_equal(got.view(np.uint32), want.reshape(-1).view(np.uint32)):
                bad.append(("ai", w, h, frames, r, int((got != want.reshape(-1)).sum())))
                break
    c.close()

def ldp_case(w, h, seed, reps):
    rng = np.random.default_rng(seed)
    cblob, lblob = oracle.synth_blob(seed, 1.0), ol.synth_lstm_blob(seed, 3.0)
    c = pkg.EthCnn(0)
    c.load_blob(cblob); c.load_lstm_blob(lblob)
    gs = os_ = None
    for i in range(1, reps + 1):
        luma = np.clip(128 + rng.laplace(0, 7, size=(h, w)), 0, 255).astype(np.uint8)
        gp, gs = c.ldp_predict_frame(luma, w, h, 32, i, gs)
        op, os_ = ol.lstm_step(lblob, oracle.resi_vectors(cblob, luma, w, h), os_, 32, i, 0.5, 0.5, mode=0)
        if not (np.array_equal(gp.view(np.uint32), op.view(np.uint32)) and np.array_equal(gs.view(np.uint32), os_.view(np.uint32))):
            bad.append(("ldp", w, h, i)); break
    c.close()

t0 = time.time()
threads = [threading.Thread(target=ai_case, args=(1920, 1080, 6, 32, 1, REPS)),
           threading.Thread(target=ai_case, args=(200, 136, 9, 22, 2, REPS * 3)),
           threading.Thread(target=ai_case, args=(4928, 3264, 1, 37, 3, REPS)),
           threading.Thread(target=ldp_case, args=(832, 480, 4, REPS // 3))]
for t in threads: t.start()
for t in threads: t.join()
ai_case(3840, 2160, 12, 27, 5, REPS // 3)
print("soak: %d reps, %.0f s, failures: %s" % (REPS, time.time() - t0, bad or "none"))
sys.exit(1 if bad else 0)

"""FC1 time for short row ranges, per tile shape (ETHCNN_FC1_VARIANT is read once per process, so
each shape runs in its own child process).  Usage: python scripts/fc1_rows.py [variants...]"""
import importlib, os, subprocess, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROWS = [int(x) for x in os.environ.get("ROWS", "256,512,924,2048,3696,8192").split(",")]
if os.environ.get("FC1_CHILD"):
    sys.path.insert(0, ROOT)
    import numpy as np
    pkg = importlib.import_module("hevc-complexity-reduction_amd")
    ctx = pkg.EthCnn(0)
    ctx.load_synthetic(1, 1.0)
    out = []
    for n in ROWS:
        w = 64 * n
        luma = np.random.default_rng(n).integers(0, 256, size=(64, w), dtype=np.uint8)
        d_in, d_vec = ctx.alloc(luma.nbytes), ctx.alloc(n * 448 * 4)
        d_in.upload(luma)
        call = lambda: ctx._chk(ctx.lib.ethcnn_resi_vectors_device(ctx.h, d_in.ptr, w, 64, w, d_vec.ptr))
        for _ in range(5): call()
        ctx.synchronize(); ctx.set_profiling(2); ctx.reset_stage_times()
        for _ in range(30): call()
        st = ctx.stage_times(); ctx.set_profiling(0)
        crc = zlib.crc32(d_vec.download(np.float32, n * 448).tobytes())
        out.append("%d:%.1f:%08x" % (n, st["ms"]["fc1"] / 30 * 1e3, crc))
        d_in.free(); d_vec.free()
    print("RES " + " ".join(out))
    sys.exit(0)
variants = sys.argv[1:] or ["-1", "0", "1", "2", "3", "4", "5", "6"]
print("rows      " + " ".join("%9d" % n for n in ROWS))
ref = None
for v in variants:
    env = dict(os.environ, FC1_CHILD="1")
    if v != "-1": env["ETHCNN_FC1_VARIANT"] = v
    r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RES ")]
    if not line:
        print("variant %s failed: %s" % (v, r.stderr[-300:])); continue
    items = [x.split(":") for x in line[0][4:].split()]
    crcs = [x[2] for x in items]
    if ref is None: ref = crcs
    print("v%-4s us  " % v + " ".join("%9s" % x[1] for x in items) + ("  results identical" if crcs == ref else "  RESULTS DIFFER"))

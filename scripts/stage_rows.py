"""Per-stage kernel time (HIP events, us) of the AI pass for a range of batch sizes (rows = CTUs in
one 64-row strip frame).  Usage: ROWS=... python scripts/stage_rows.py"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
pkg = importlib.import_module("hevc-complexity-reduction_amd")
ROWS = [int(x) for x in os.environ.get("ROWS", "64,1024,4096,8192,16384,24576,32768,49152,65536").split(",")]
ctx = pkg.EthCnn(0)
ctx.load_synthetic(1, 1.0)
print("%8s %8s %8s %8s %8s %8s" % ("rows", "tile", "trunk", "fc1", "heads", "gate"))
for n in ROWS:
    w = 64 * n
    luma = np.random.default_rng(n).integers(0, 256, size=(64, w), dtype=np.uint8)
    d_in, d_out = ctx.alloc(luma.nbytes), ctx.alloc(n * 84)
    d_in.upload(luma)
    for _ in range(3): ctx.predict_luma_device(d_in, w, 64, 1, 32, d_out)
    ctx.synchronize(); ctx.set_profiling(2); ctx.reset_stage_times()
    R = 20
    for _ in range(R): ctx.predict_luma_device(d_in, w, 64, 1, 32, d_out)
    st = ctx.stage_times()["ms"]; ctx.set_profiling(0)
    print("%8d " % n + " ".join("%8.1f" % (st[k] / R * 1e3) for k in ("tile", "trunk", "fc1", "heads", "gate")))
    d_in.free(); d_out.free()

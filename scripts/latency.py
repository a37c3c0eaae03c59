"""Small-batch latency of the device entry points (inputs resident in HBM, one call + sync)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
pkg = importlib.import_module("hevc-complexity-reduction_amd")
ctx = pkg.EthCnn(0)
ctx.load_synthetic(1, 8.0)
rows = []
for name, w, h, frames in (("c1 768x512x1", 768, 512, 1), ("1080p x1", 1920, 1080, 1), ("2160p x1", 3840, 2160, 1), ("4928x3264 x1", 4928, 3264, 1)):
    luma = bench.synth_luma(w, h, frames, 3)
    nctu = pkg.ethcnn.ctus_per_frame(w, h)
    d_in, d_out, d_vec = ctx.alloc(luma.nbytes), ctx.alloc(frames * nctu * 84), ctx.alloc(nctu * 448 * 4)
    d_in.upload(luma)
    for kind in ("predict", "resi"):
        def call():
            if kind == "predict":
                ctx.predict_luma_device(d_in, w, h, frames, 32, d_out)
            else:
                ctx._chk(ctx.lib.ethcnn_resi_vectors_device(ctx.h, d_in.ptr, w, h, w, d_vec.ptr))
            ctx.synchronize()
        for _ in range(20): call()
        t0 = time.perf_counter(); n = 200
        for _ in range(n): call()
        dt = (time.perf_counter() - t0) / n
        rows.append("%-14s %-8s %5d CTUs  %8.1f us/call  %8.2f M CTU/s" % (name, kind, frames * nctu, dt * 1e6, frames * nctu / dt / 1e6))
    d_in.free(); d_out.free(); d_vec.free()
print("\n".join(rows))

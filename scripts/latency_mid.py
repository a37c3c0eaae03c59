"""Single-picture latency between 1080p and 2160p (device-resident, launch + completion word); ETHCNN_SMALL_SHAPE=0/1 forces the
FC1 / heads form of the single-launch pass (0: register-fed 64 x 16, 1: 64 x 32)."""
import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
pkg = importlib.import_module("hevc-complexity-reduction_amd")
ctx = pkg.EthCnn(0)
ctx.load_synthetic(1, 8.0)
for name, w, h in (("1600x1216", 1600, 1216), ("2048x1152", 2048, 1152), ("2560x1440", 2560, 1440), ("2560x2048", 2560, 2048), ("3072x2048", 3072, 2048), ("3840x1920", 3840, 1920), ("3840x2160", 3840, 2160)):
    luma = bench.synth_luma(w, h, 1, 3)
    nctu = pkg.ethcnn.ctus_per_frame(w, h)
    d_in, d_out, d_vec = ctx.alloc(luma.nbytes), ctx.alloc(nctu * 84), ctx.alloc(nctu * 448 * 4)
    d_in.upload(luma)
    for kind in ("predict", "resi"):
        def call():
            if kind == "predict": ctx.predict_luma_device(d_in, w, h, 1, 32, d_out)
            else: ctx.resi_vectors_device(d_in, w, h, d_vec)
            ctx.synchronize()
        for _ in range(50): call()
        t0 = time.perf_counter(); n = 300
        for _ in range(n): call()
        print("%-10s %-8s %5d CTUs %7.1f us" % (name, kind, nctu, (time.perf_counter() - t0) / n * 1e6))

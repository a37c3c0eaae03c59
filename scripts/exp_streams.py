"""Experiment: does running sub-batches on several streams (contexts) raise whole-batch throughput?"""
import importlib, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
pkg = importlib.import_module("hevc-complexity-reduction_amd")
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
W, H, NF, QP = wl["width"], wl["height"], wl["frames"], wl["qp"]
nctu = pkg.ethcnn.ctus_per_frame(W, H)
luma = bench.synth_luma(W, H, NF, 1)
for nsplit in (1, 2, 3, 4):
    ctxs = [pkg.EthCnn(0) for _ in range(nsplit)]
    parts = []
    for i, c in enumerate(ctxs):
        c.load_synthetic(1, 8.0)
        f0, f1 = i * NF // nsplit, (i + 1) * NF // nsplit
        d_in = c.alloc(luma[f0:f1].nbytes); d_out = c.alloc((f1 - f0) * nctu * 84)
        d_in.upload(luma[f0:f1]); parts.append((c, d_in, d_out, f1 - f0))
    def step():
        for c, di, do, nf in parts:
            c.predict_luma_device(di, W, H, nf, QP, do)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K): step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("streams=%d: %.2f M CTU/s (%.3f ms/step)" % (nsplit, NF * nctu * K / dt / 1e6, dt / K * 1e3))
    for c, di, do, nf in parts:
        di.free(); do.free(); c.close()

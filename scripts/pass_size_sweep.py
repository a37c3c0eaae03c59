#!/usr/bin/env python
"""GPU box: C3 (3840x2160 x 50 frames) per-step time against the pass size (ethcnn_options.max_ctus_per_pass), per plan.
A pass of n CTUs keeps n x 10,752 B of feature pieces (plans 2 / 3; 10,752 B of fp32 features in plan 0) between the trunk and
FC1: the whole step in one pass is 1.1 GB through HBM both ways; a pass of 16 k CTUs is 176 MB -- inside the 256 MB memory-side cache?"""
import importlib
import sys
import time

sys.path.insert(0, ".")
import bench  # noqa: E402

pkg = importlib.import_module("hevc-complexity-reduction_amd")
W, H, NF, QP = 3840, 2160, 50, 32
luma = bench.synth_luma(W, H, NF, seed=0xE7C00002)
nctu = pkg.ethcnn.ctus_per_frame(W, H)
plans = [int(a) for a in sys.argv[1:]] or [0, 3]
for plan in plans:
    for frames_per_pass in (2, 4, 8, 16, 25, 50):
        ctx = pkg.EthCnn(device=0, max_ctus_per_pass=frames_per_pass * nctu)
        ctx.load_synthetic(seed=1, head_gain=8.0)
        ctx.set_thresholds(0.5, 0.5)
        ctx.set_fc1_plan(plan)
        d_in = ctx.alloc(luma.nbytes)
        d_out = ctx.alloc(nctu * NF * 21 * 4)
        d_in.upload(luma)
        t = time.perf_counter()
        while time.perf_counter() - t < 0.3:
            ctx.predict_luma_device(d_in, W, H, NF, QP, d_out)
            ctx.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(20):
                ctx.predict_luma_device(d_in, W, H, NF, QP, d_out)
            ctx.synchronize()
            best = min(best, (time.perf_counter() - t0) / 20)
        print("plan %d  %2d frames = %6d CTUs per pass (%4.0f MB between trunk and FC1): %.3f ms per step  %.2f M CTU/s" %
              (plan, frames_per_pass, frames_per_pass * nctu, frames_per_pass * nctu * 10752 / 1e6, best * 1e3, nctu * NF / best / 1e6), flush=True)
        ctx.close()

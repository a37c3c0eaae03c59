"""Streamed input of the LDP step (ethcnn_ldp_step_begin / rows_ready / end) against the plain call, page-locked buffers, the filling
done in-process by a small C-speed loop (numpy row copies out of a second buffer by 1, 2, 4 threads):
  plain        fill the buffer (same copy), then ethcnn_ldp_step
  streamed     ethcnn_ldp_step_begin, then fill CTU row by CTU row reporting each, then ethcnn_ldp_step_end
  no-wait      every row reported before begin (what the streamed launch costs when nothing has to be waited for)"""
import importlib, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
pkg = importlib.import_module("hevc-complexity-reduction_amd")
e = pkg.ethcnn
ctx = pkg.EthCnn(0)
ctx.load_synthetic(1, 1.0)
ctx.load_lstm_checkpoint(os.path.join(ROOT, "tests", "golden", "model_LDP_200000_qp32.dat"))
rng = np.random.default_rng(0)
reps = 200
for name, w, h in (("832x480", 832, 480), ("1280x720", 1280, 720), ("1920x1080", 1920, 1080), ("3840x2160", 3840, 2160)):
    n, nrows = e.ctus_per_frame(w, h), (h + 63) // 64
    src = np.clip(128 + rng.laplace(0, 6, size=(h, w)), 0, 255).astype(np.uint8).reshape(-1)
    pin = ctx.host_buffer(w * h)
    pprobs = ctx.host_buffer(n * 84).view(np.float32)
    def fill(cy0, cy1, report):
        for cy in range(cy0, cy1):
            a, b = cy * 64 * w, min(h, cy * 64 + 64) * w
            pin[a:b] = src[a:b]
            if report:
                ctx.lib.ethcnn_rows_ready(ctx.h, cy, cy + 1)
    res = {}
    for mode in ("plain", "streamed", "no-wait"):
        for i in range(1, 6 + reps):
            if i == 6:
                t0 = time.perf_counter()
            if mode == "plain":
                fill(0, nrows, False)
                ctx.ldp_step(pin, w, h, 32, i, probs_out=pprobs)
            elif mode == "streamed":
                ctx.ldp_step_begin(pin, w, h, 32, i, pprobs)
                fill(0, nrows, True)
                ctx.ldp_step_end()
            else:
                fill(0, nrows, False)
                ctx.lib.ethcnn_rows_ready(ctx.h, 0, nrows)
                ctx.ldp_step_begin(pin, w, h, 32, i, pprobs)
                ctx.ldp_step_end()
        res[mode] = (time.perf_counter() - t0) / reps * 1e6
    t0 = time.perf_counter()
    for _ in range(reps):
        fill(0, nrows, False)
    t_fill = (time.perf_counter() - t0) / reps * 1e6
    ctx.free_host_buffers()
    print("%-10s %5d CTUs  filling alone %6.1f us (one thread, numpy row copies) | fill, then ethcnn_ldp_step %6.1f us | begin, fill + report, end %6.1f us | fill, report all, begin, end %6.1f us"
          % (name, n, t_fill, res["plain"], res["streamed"], res["no-wait"]))

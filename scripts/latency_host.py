"""Single-picture latency of the HOST entry point (pageable luma -> probabilities in host memory): what the in-process HM hook
pays per picture (tools/hm_inprocess_hook.c -> ethcnn_predict_luma)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
pkg = importlib.import_module("hevc-complexity-reduction_amd")
ctx = pkg.EthCnn(0)
bench.pin_to_gpu_numa_node(ctx.device_name)
ctx.load_synthetic(1, 8.0)
for name, w, h in (("416x240", 416, 240), ("768x512", 768, 512), ("1280x720", 1280, 720), ("1920x1080", 1920, 1080), ("3840x2160", 3840, 2160), ("4928x3264", 4928, 3264)):
    luma = bench.synth_luma(w, h, 1, 3)
    nctu = pkg.ethcnn.ctus_per_frame(w, h)
    for _ in range(20):
        ctx.predict_luma(luma, w, h, 1, 32)
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.predict_luma(luma, w, h, 1, 32)
    dt = (time.perf_counter() - t0) / n
    # the same from page-locked buffers (ethcnn_host_alloc), as tools/hm_inprocess_hook.c holds its 8-bit copy of HM's picture
    pin = ctx.host_buffer(w * h)
    pin[:] = luma.reshape(-1)
    pout = ctx.host_buffer(nctu * 84).view(np.float32)
    import ctypes
    def call():
        ctx._chk(ctx.lib.ethcnn_predict_luma(ctx.h, pin.ctypes.data, w, h, w, w * h, 1, 32, pout.ctypes.data_as(ctypes.POINTER(ctypes.c_float))))
    for _ in range(20):
        call()
    t0 = time.perf_counter()
    for _ in range(n):
        call()
    dtp = (time.perf_counter() - t0) / n
    ctx.free_host_buffers()
    print("%-10s %5d CTUs  predict_luma host -> host: pageable %7.1f us/picture, page-locked %7.1f us  (%.1f MB of luma = %.1f us at 55 GB/s)"
          % (name, nctu, dt * 1e6, dtp * 1e6, w * h / 1e6, w * h / 55e3))

"""What the in-process encoder hook (tools/hm_inprocess_hook.c, ethcnn_hm_predict_picture) costs HM per picture: the conversion of
its 16-bit picture to 8 bits + the prediction -- streamed (the pass is queued first and takes the CTU rows as the conversion loop
produces them: ethcnn_predict_luma_begin / ethcnn_rows_ready / ethcnn_predict_luma_end) against convert-then-predict
(ETHCNN_HM_STREAM=0).  The hook is compiled here as a shared object (gcc -O2, as oracle/build_ref_hm.sh does) and called through ctypes,
one process per setting (the switch is read once)."""
import ctypes, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "hevc-complexity-reduction_amd", "lib")

def child(so, stream):
    import numpy as np
    os.environ["ETHCNN_HM_STREAM"] = stream
    os.environ["ETHCNN_SYNTHETIC_SEED"] = "5"
    d = tempfile.mkdtemp()
    os.chdir(d)
    open("Thr_info.txt", "w").write("0.5 0.5 0.5 0.5 0.5 0.5\n")
    hook = ctypes.CDLL(so)
    fn = hook.ethcnn_hm_predict_picture
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    rng = np.random.default_rng(1)
    out = []
    for (w, h) in ((416, 240), (1280, 720), (1920, 1080), (3840, 2160)):
        stride = w + 160  # HM's picture buffers carry margins
        pic = rng.integers(0, 256, size=(h, stride), dtype=np.int16)
        n = ((w + 63) // 64) * ((h + 63) // 64)
        probs = np.empty(n * 21, dtype=np.float32)
        for _ in range(10):
            assert fn(pic.ctypes.data, stride, w, h, 8, 32, probs.ctypes.data) == 0
        reps = 200
        t0 = time.perf_counter()
        for _ in range(reps):
            fn(pic.ctypes.data, stride, w, h, 8, 32, probs.ctypes.data)
        out.append("%dx%d %.1f %08x" % (w, h, (time.perf_counter() - t0) / reps * 1e6, int(np.frombuffer(probs.tobytes(), dtype=np.uint32).sum() & 0xffffffff)))
    print(" | ".join(out))

if __name__ == "__main__":
    if len(sys.argv) > 2:
        child(sys.argv[1], sys.argv[2])
        sys.exit(0)
    so = os.path.join(tempfile.mkdtemp(), "libhook.so")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-D_POSIX_C_SOURCE=200809L", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "hm_inprocess_hook.c"), "-o", so, "-L" + LIB, "-lethcnn", "-Wl,-rpath," + LIB, "-Wl,-rpath,/opt/rocm/lib"])
    print("# ethcnn_hm_predict_picture (16-bit picture with HM's margins -> 8 bits -> probabilities), us per picture, 200 pictures; last field: checksum of the probabilities")
    for stream, label in (("1", "streamed (as shipped)"), ("0", "ETHCNN_HM_STREAM=0: convert, then ethcnn_predict_luma")):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), so, stream], capture_output=True, text=True, timeout=600)
        print("%-55s %s" % (label, r.stdout.strip() if r.returncode == 0 else "FAILED " + r.stderr[-400:]))

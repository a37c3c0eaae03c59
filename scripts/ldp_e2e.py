#!/usr/bin/env python
"""End-to-end Low-Delay-P run with the reference's REAL encoder side of the handshake:
oracle/_ref/hm_ldp/TAppEncoderLDP (HM-16.5_Test_LDP built unchanged by oracle/build_ref_hm.sh) encodes a
synthetic moving sequence while a predictor daemon answers its command.dat / pred_start.sig requests.

    ldp_e2e.py gpu    <outdir>   daemon = hevc-complexity-reduction_amd/resi_to_cu_depth_LDP.serve (MI355X)
    ldp_e2e.py gpu-cli <outdir>  started as a separate process through the root launcher resi_to_cu_depth_LDP.py (default = C daemon)
    ldp_e2e.py gpu-cli-python <outdir>  the launcher with --python (the Python daemon)
    ldp_e2e.py gpu-native <outdir>  the native daemon (tools/resi_to_cu_depth_ldp.c over the C ABI), a separate process
    ldp_e2e.py oracle <outdir>   daemon = the same protocol answered by the CPU oracle (test infrastructure)

Both write <outdir>/<mode>.json: bitstream md5 + a crc32 of every frame's cu_depth.dat / state.dat.
The two must agree exactly (bit-exact predictions -> identical encoder decisions -> identical bitstream).
Weights: the reference's trained LSTM checkpoint (qp 32 band, tests/golden) + seeded synthetic CNN
weights (the trained LDP CNN blob is not in the reference repository).
"""
import hashlib
import importlib
import json
import os
import shutil
import subprocess
import sys
import threading
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
W, H, FRAMES, QP, SEED = 416, 240, 6, 32, 21
GOLD = os.path.join(ROOT, "tests", "golden", "model_LDP_200000_qp32.dat")


def make_sequence(path):
    """a textured scene panning 3 px / frame with a little noise: real motion-compensated residuals"""
    rng = np.random.default_rng(4)
    big = rng.integers(0, 256, size=(H + 64, W + 128)).astype(np.float64)
    for _ in range(3):
        big = (big + np.roll(big, 1, 0) + np.roll(big, 1, 1) + np.roll(np.roll(big, 1, 0), 1, 1)) / 4
    big = (big - big.min()) / (big.max() - big.min()) * 200 + 25
    yy, xx = np.mgrid[0:H + 64, 0:W + 128]
    big += 20 * np.sin(xx / 9.0) * np.cos(yy / 7.0)
    with open(path, "wb") as f:
        for k in range(FRAMES):
            fr = big[2 * k:2 * k + H, 3 * k:3 * k + W] + rng.normal(0, 1.5, size=(H, W))
            f.write(np.clip(fr, 0, 255).astype(np.uint8).tobytes())
            f.write(np.full(W * H // 2, 128, np.uint8).tobytes())


def oracle_daemon(workdir, max_frames, log):
    import ethcnn_np as oracle
    import ethcnn_lstm_np as ol
    cblob = oracle.synth_blob(SEED, 1.0)
    lblob = np.fromfile(GOLD + ".data-00000-of-00001", dtype=np.float32)
    thr = [float(t) for t in open(os.path.join(workdir, "Thr_info.txt")).readline().split(" ")]
    p = lambda n: os.path.join(workdir, n)
    n_done = 0
    while n_done < max_frames:
        if not os.path.isfile(p("pred_start.sig")):
            time.sleep(2e-4)
            continue
        parts = open(p("command.dat")).readline().split(" ")
        if len(parts) != 5 or parts[4] != "[end]":
            continue
        i_frame, w, h, qp = (int(x) for x in parts[:4])
        os.remove(p("pred_start.sig"))
        luma = np.frombuffer(open(p("resi.yuv"), "rb").read(w * h), dtype=np.uint8).reshape(h, w)
        n = ((w + 63) // 64) * ((h + 63) // 64)
        state = np.fromfile(p("state.dat"), dtype=np.float32).reshape(n, 2, 448) if i_frame > 1 else None
        vec = oracle.resi_vectors(cblob, luma, w, h)
        probs, st = ol.lstm_step(lblob, vec, state, qp, i_frame, thr[1], thr[3], mode=0)
        st.astype(np.float32).tofile(p("state.dat"))
        probs.astype(np.float32).tofile(p("cu_depth.dat"))
        log.append((i_frame, zlib.crc32(probs.tobytes()), zlib.crc32(st.tobytes())))
        open(p("pred_end.sig"), "wb").close()
        n_done += 1


def gpu_daemon(workdir, max_frames, log):
    os.environ["ETHCNN_SYNTHETIC_SEED"] = str(SEED)
    pkg = importlib.import_module("hevc-complexity-reduction_amd")
    d = pkg.resi_to_cu_depth_LDP
    orig = d.save_cu_depth_and_state

    def logging_save(depth_out, state_out, save_file, state_file, end_file, num_vectors, **kw):
        frame = len(log) + 1
        crc_p = zlib.crc32(np.ascontiguousarray(depth_out, np.float32).tobytes())  # before the signal: the buffer is reused
        state = orig(depth_out, state_out, save_file, state_file, end_file, num_vectors, **kw)  # fetches the resident state after the signal
        log.append((frame, crc_p, zlib.crc32(np.ascontiguousarray(state, np.float32).tobytes())))
        return state
    d.save_cu_depth_and_state = logging_save
    d.serve(workdir, max_frames=max_frames, idle_timeout=600.0, verbose=False)


def main():
    mode, out = sys.argv[1], os.path.abspath(sys.argv[2])
    work = os.path.join(out, "work_" + mode)
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(work)
    make_sequence(os.path.join(work, "seq.yuv"))
    open(os.path.join(work, "Thr_info.txt"), "w").write("0.4 0.6 0.3 0.7 0.2 0.8")  # as shipped in HM-16.5_Test_LDP/bin
    for ext in (".index", ".data-00000-of-00001"):
        shutil.copy(GOLD + ext, os.path.join(work, "model_LDP_200000_qp32.dat" + ext))
    log = []
    cli = None
    if mode in ("gpu-cli", "gpu-cli-python"):
        # the daemon exactly as a user starts it: `python resi_to_cu_depth_LDP.py` in the encoder's directory (default: the launcher
        # execs the C daemon when it is built; gpu-cli-python: `--python`, the Python daemon)
        os.symlink(os.path.join(ROOT, "resi_to_cu_depth_LDP.py"), os.path.join(work, "resi_to_cu_depth_LDP.py"))
        cli = subprocess.Popen([sys.executable, "resi_to_cu_depth_LDP.py", "--max-frames", str(FRAMES - 1), "--idle-timeout", "300"]
                               + (["--python"] if mode == "gpu-cli-python" else []),
                               cwd=work, env=dict(os.environ, ETHCNN_SYNTHETIC_SEED=str(SEED)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        th = threading.Thread(target=cli.wait, daemon=True)
    elif mode == "gpu-native":  # the C daemon over the ABI, started in the encoder's directory like the Python one
        exe_d = os.path.join(ROOT, "hevc-complexity-reduction_amd", "bin", "resi_to_cu_depth_ldp")
        cli = subprocess.Popen([exe_d, "--max-frames", str(FRAMES - 1), "--idle-timeout", "300"],
                               cwd=work, env=dict(os.environ, ETHCNN_SYNTHETIC_SEED=str(SEED)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        th = threading.Thread(target=cli.wait, daemon=True)
    else:
        th = threading.Thread(target=(gpu_daemon if mode == "gpu" else oracle_daemon), args=(work, FRAMES - 1, log), daemon=True)
    th.start()
    time.sleep(6.0 if mode.startswith("gpu-cli") else (3.0 if mode in ("gpu", "gpu-native") else 0.5))  # the reference's daemon is started by hand before the encoder, too
    exe = os.path.join(ROOT, "oracle", "_ref", "hm_ldp", "TAppEncoderLDP")
    t0 = time.time()
    r = subprocess.run([exe, "-c", os.path.join(ROOT, "scripts", "hm_ldp_test.cfg"), "-i", "seq.yuv", "-wdt", str(W), "-hgt", str(H),
                        "-fr", "30", "-f", str(FRAMES), "-q", str(QP), "-b", "str.bin", "-o", ""],
                       cwd=work, capture_output=True, text=True, timeout=200)
    enc_s = time.time() - t0
    if r.returncode != 0:
        raise SystemExit("HM-LDP failed:\n" + r.stdout[-2000:] + r.stderr[-2000:])
    th.join(timeout=30)
    pred = [l.strip() for l in r.stdout.splitlines() if "Predicting Time" in l]
    res = {"mode": mode, "frames": FRAMES, "bitstream_md5": hashlib.md5(open(os.path.join(work, "str.bin"), "rb").read()).hexdigest(),
           "bitstream_bytes": os.path.getsize(os.path.join(work, "str.bin")), "per_frame_crc": log,
           "encoder_seconds": enc_s, "hm_predicting_time_lines": pred,
           # what the daemon left behind: the last frame's files (every mode must agree on them byte for byte)
           "final_cu_depth_crc": zlib.crc32(open(os.path.join(work, "cu_depth.dat"), "rb").read()),
           "final_state_crc": zlib.crc32(open(os.path.join(work, "state.dat"), "rb").read())}
    json.dump(res, open(os.path.join(out, mode + ".json"), "w"), indent=1)
    print(json.dumps(res))
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()

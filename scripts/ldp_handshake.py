#!/usr/bin/env python
"""Config #5 where the ENCODER stands: tools/ldp_client.c plays HM's side of the file handshake exactly
as TEncGOP.cpp:1466-1506 does (command.dat, pred_start.sig, busy wait on pred_end.sig, fread cu_depth.dat) against

    python   the Python daemon   (hevc-complexity-reduction_amd/resi_to_cu_depth_LDP.py through the root launcher)
    native   the C daemon        (tools/resi_to_cu_depth_ldp.c over the C ABI; inotify wake-up; resi.yuv streamed into the running
             prediction: ethcnn_ldp_step_begin / rows_ready / end;  native-spin: + --spin, the reference daemon's busy wait for
             pred_start.sig;  native-nostream: --no-stream, read first, then predict)

for 1920x1080 and 416x240, frames back to back and with a 5 ms gap (an encoder encodes between two requests), working directory
on tmpfs (/dev/shm) and on the box's disk (/tmp).  Per run: handshake p50 / p90 / p99 in us, and that the two daemons answered
every frame with byte-identical cu_depth.dat (per-frame digests) and left byte-identical state.dat behind.

    python scripts/ldp_handshake.py [frames]      (GPU box)  -> stdout (profiles/r04_ldp_handshake.txt)
"""
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "hevc-complexity-reduction_amd", "bin")
GOLD = os.path.join(ROOT, "tests", "golden", "model_LDP_200000_qp32.dat")
SEED = "21"


def run(kind, base, w, h, frames, gap_us, qp=32):
    work = tempfile.mkdtemp(prefix="ldp_%s_" % kind, dir=base)
    try:
        open(os.path.join(work, "Thr_info.txt"), "w").write("0.4 0.6 0.3 0.7 0.2 0.8")
        for ext in (".index", ".data-00000-of-00001"):
            shutil.copy(GOLD + ext, os.path.join(work, "model_LDP_200000_qp32.dat" + ext))
        env = dict(os.environ, ETHCNN_SYNTHETIC_SEED=SEED)
        if kind == "python":
            os.symlink(os.path.join(ROOT, "resi_to_cu_depth_LDP.py"), os.path.join(work, "resi_to_cu_depth_LDP.py"))
            cmd = [sys.executable, "resi_to_cu_depth_LDP.py", "--python", "--max-frames", str(frames), "--idle-timeout", "120"]
        else:
            cmd = [os.path.join(BIN, "resi_to_cu_depth_ldp"), "--max-frames", str(frames), "--idle-timeout", "120", "--quiet", "--trace"]
            if kind == "native-nostream":  # A/B: read resi.yuv first, then predict (the round's first form of the daemon)
                cmd.append("--no-stream")
            if kind == "native-spin":      # busy-wait for pred_start.sig, as the reference's daemon does
                cmd.append("--spin")
        d = subprocess.Popen(cmd, cwd=work, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        time.sleep(6.0 if kind == "python" else 2.5)  # the daemon is started before the encoder, as with the reference
        c = subprocess.run([os.path.join(BIN, "ldp_client"), work, str(w), str(h), str(qp), str(frames), "--gap-us", str(gap_us),
                            "--digest", os.path.join(work, "digest.txt")], capture_output=True, text=True, timeout=600)
        try:
            d.wait(timeout=60)
        except subprocess.TimeoutExpired:
            d.kill()
        if c.returncode != 0 or d.returncode != 0:
            raise SystemExit("%s daemon run failed: client %d %s | daemon %s %s" % (kind, c.returncode, c.stderr[-400:], d.returncode, d.stderr.read()[-400:]))
        digest = open(os.path.join(work, "digest.txt")).read()
        state = hashlib.md5(open(os.path.join(work, "state.dat"), "rb").read()).hexdigest()
        tr = [l for l in d.stderr.read().splitlines() if l.startswith("trace")]
        return c.stdout.strip() + ("\n         " + tr[0] if tr else ""), digest, state
    finally:
        shutil.rmtree(work, ignore_errors=True)


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    print("# LDP handshake from the encoder's side: tools/ldp_client.c (HM's sequence of file operations), %d frames per run" % frames)
    print("# library call alone (ethcnn_ldp_step, page-locked buffers): profiles/r04_latency_ldp.txt")
    ok = True
    for base in ("/dev/shm", "/tmp"):
        if not os.path.isdir(base):
            continue
        for (w, h) in ((1920, 1080), (416, 240)):
            for gap in (0, 5000):
                res = {}
                for kind in ("python", "native", "native-spin", "native-nostream"):
                    line, digest, state = run(kind, base, w, h, frames if gap == 0 else max(100, frames // 5), gap)
                    res[kind] = (digest, state)
                    print("%-8s %-15s %s" % (base, kind, line))
                same = res["python"] == res["native"] == res["native-nostream"] == res["native-spin"]
                ok &= same
                print("         -> per-frame cu_depth.dat digests and the final state.dat of the four runs: %s" % ("IDENTICAL" if same else "DIFFER"))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())

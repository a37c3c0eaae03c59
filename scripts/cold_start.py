"""Cold start and fixed cost of the drop-in command: the encoder blocks in system() on
`python video_to_cu_depth.py <yuv> <w> <h> <qp>` (TAppEncCfg.cpp:2317-2321), so what it sees is the WALL time of the command --
process start, imports, HIP runtime init, context, checkpoint, first launch, compute, exit.  The reference quotes "1~10 s" for its
TensorFlow start-up (README.md:124).  Measured here per config and per launcher, 1 / 2 / 8 workers on the one visible GPU:

    python scripts/cold_start.py [reps]      -> gpurun_out/cold_start.txt  (profiles/r06_cold_start.txt)

C1 768x512 x 1 (96 CTUs), C2 1920x1080 x 50, C4's 1/8 share 4928x3264 x 54.  Launchers: the Python host mirror (one process; workers =
threads inside the library), the same with a PROCESS per worker (ETHCNN_SHARD_PROCESSES=1, round 5's form), the native C tool.  Every
output is compared byte for byte with the first one of its config."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
TOOL = os.path.join(ROOT, "hevc-complexity-reduction_amd", "bin", "video_to_cu_depth")
CONFIGS = [("C1 768x512 x 1", 768, 512, 1), ("C2 1920x1080 x 50", 1920, 1080, 50), ("C4/8 4928x3264 x 54", 4928, 3264, 54)]
WORKERS = ["0", "0,0", ",".join(["0"] * 8)]


def run(cmd, cwd, devices, extra=None):
    env = dict(os.environ, ETHCNN_SYNTHETIC_SEED="3", ETHCNN_HEAD_GAIN="8", ETHCNN_DEVICES=devices, ETHCNN_TIMING="1")
    env.update(extra or {})
    env["ETHCNN_T0_MS"] = "%.3f" % (time.perf_counter() * 1e3)
    t0 = time.perf_counter()
    r = subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True)
    wall = (time.perf_counter() - t0) * 1e3
    if r.returncode != 0:
        raise SystemExit("%s failed: %s" % (cmd, r.stderr[-600:]))
    return wall, [l for l in r.stderr.splitlines() if "timing" in l]


def main():
    out = []
    d = tempfile.mkdtemp(prefix="cold_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    open(os.path.join(d, "Thr_info.txt"), "w").write("0.5 0.5 0.5 0.5 0.5 0.5\n")
    rng = np.random.default_rng(1)
    launchers = [("python launcher, worker threads", [sys.executable, os.path.join(ROOT, "video_to_cu_depth.py")], None),
                 ("python launcher, worker processes", [sys.executable, os.path.join(ROOT, "video_to_cu_depth.py")], {"ETHCNN_SHARD_PROCESSES": "1"}),
                 ("native C tool", [TOOL], None)]
    for name, w, h, nf in CONFIGS:
        yuv = os.path.join(d, "seq.yuv")
        frame = rng.integers(0, 256, size=w * h * 3 // 2, dtype=np.uint8)
        with open(yuv, "wb") as f:
            for k in range(nf):
                f.write(np.roll(frame, 977 * k).tobytes())
        nctu = ((w + 63) // 64) * ((h + 63) // 64)
        out.append("== %s: %d CTUs, %.0f MB file (page cache / tmpfs)" % (name, nctu * nf, os.path.getsize(yuv) / 1e6))
        ref = None
        for lname, cmd, extra in launchers:
            for devs in WORKERS:
                if extra and devs == "0":
                    continue  # (one worker: the same command as the thread form)
                walls, lines = [], []
                for rep in range(REPS + 1):
                    wall, lines = run(cmd + ["seq.yuv", str(w), str(h), "32"], d, devs, extra)
                    if rep:  # the first run of a setting warms the page cache / code pages
                        walls.append(wall)
                    got = open(os.path.join(d, "cu_depth.dat"), "rb").read()
                    ref = ref or got
                    assert got == ref and len(got) == nctu * nf * 84, (name, lname, devs)
                walls.sort()
                out.append("  %-34s %d worker%s: wall %8.1f ms (median of %d; min %.1f max %.1f)" % (lname, devs.count(",") + 1, " " if devs == "0" else "s",
                                                                                                    walls[len(walls) // 2], REPS, walls[0], walls[-1]))
                for l in lines:
                    out.append("      " + l)
        os.remove(yuv)
    text = "\n".join(out)
    print(text)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "cold_start.txt"), "w").write(text + "\n")


if __name__ == "__main__":
    main()

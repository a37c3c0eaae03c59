#!/usr/bin/env python
"""Where the native LDP daemon's tail latency on tmpfs comes from: tools/ldp_client.c with --slow-us against
tools/resi_to_cu_depth_ldp with --trace-slow, same CLOCK_MONOTONIC on both sides, so a slow handshake can be split into
    client: remove + command.dat | create pred_start.sig | WAIT | remove pred_end.sig + read cu_depth.dat
    daemon: detection -> ending signal (its own stages), and -- from the absolute stamps -- how long after pred_start.sig was created
            the daemon saw it (wake-up) and how long after the ending signal the client saw it.
usage: python scripts/ldp_tail.py [frames]   (GPU box) -> stdout (profiles/r05_ldp_tail.txt)"""
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "hevc-complexity-reduction_amd", "bin")
GOLD = os.path.join(ROOT, "tests", "golden", "model_LDP_200000_qp32.dat")


def run(base, w, h, frames, gap_us, extra=(), client_extra=()):
    work = tempfile.mkdtemp(prefix="ldp_tail_", dir=base)
    try:
        open(os.path.join(work, "Thr_info.txt"), "w").write("0.4 0.6 0.3 0.7 0.2 0.8")
        for ext in (".index", ".data-00000-of-00001"):
            shutil.copy(GOLD + ext, os.path.join(work, "model_LDP_200000_qp32.dat" + ext))
        env = dict(os.environ, ETHCNN_SYNTHETIC_SEED="21")
        d = subprocess.Popen([os.path.join(BIN, "resi_to_cu_depth_ldp"), "--max-frames", str(frames), "--idle-timeout", "120", "--quiet", "--trace",
                              "--trace-slow", "600"] + list(extra), cwd=work, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        time.sleep(2.5)
        c = subprocess.run([os.path.join(BIN, "ldp_client"), work, str(w), str(h), "32", str(frames), "--gap-us", str(gap_us), "--slow-us", "1000"] + list(client_extra),
                           capture_output=True, text=True, timeout=900)
        try:
            d.wait(timeout=60)
        except subprocess.TimeoutExpired:
            d.kill()
        derr = d.stderr.read()
        print("%-8s %dx%d gap %d %s%s: %s" % (base, w, h, gap_us, " ".join(extra), (" client " + " ".join(client_extra)) if client_extra else "", c.stdout.strip()))
        slow_c = {int(m.group(1)): m for m in re.finditer(r"slow POC (\d+): handshake (\d+) us = remove \+ command.dat (\d+) \| create pred_start.sig (\d+) \| wait for pred_end.sig (\d+) \| "
                                                           r"remove it \+ read cu_depth.dat (\d+) ; monotonic us: pred_start.sig created (\d+), pred_end.sig seen (\d+)", c.stderr)}
        slow_d = {int(m.group(1)): m for m in re.finditer(r"slow frame (\d+): (\d+) us from detection.*?monotonic us: detected (\d+), ending signal (\d+)", derr)}
        print("    client: %d handshakes > 1000 us; daemon: %d frames > 600 us of its own" % (len(slow_c), len(slow_d)))
        shown = 0
        for poc, m in sorted(slow_c.items()):
            hs, cmd, cre, wait, rd, t_cre, t_seen = (int(m.group(i)) for i in range(2, 9))
            dm = slow_d.get(poc)
            if dm:
                d_own, t_det, t_end = int(dm.group(2)), int(dm.group(3)), int(dm.group(4))
                where = "daemon saw pred_start.sig %d us after its creation, took %d us itself, client saw pred_end.sig %d us after it was made" % (t_det - t_cre, d_own, t_seen - t_end)
            else:
                where = "the daemon's own time for this frame was < 600 us: the wait is wake-up / directory-lookup latency outside its stages"
            if shown < 12:
                print("    POC %4d: %5d us = cmd %d | create %d | wait %d | read %d -- %s" % (poc, hs, cmd, cre, wait, rd, where))
            shown += 1
        for l in derr.splitlines():
            if l.startswith("trace") or (l.startswith("slow frame") and shown < 0):
                print("    " + l[:400])
        dl = [l for l in derr.splitlines() if l.startswith("slow frame")]
        for l in dl[:8]:
            print("    daemon " + l[:330])
    finally:
        shutil.rmtree(work, ignore_errors=True)


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    for base in ("/dev/shm", "/tmp"):
        if os.path.isdir(base):
            for (w, h, gap) in ((1920, 1080, 0), (416, 240, 0), (1920, 1080, 5000)):
                run(base, w, h, frames if gap == 0 else frames // 5, gap)
    run("/dev/shm", 1920, 1080, frames, 0, ("--spin",))
    run("/dev/shm", 1920, 1080, frames, 0, ("--no-stream",))
    # round 6: is it the ENCODER's wait -- a tight loop of failing fopen("pred_end.sig") calls, TEncGOP.cpp:1483 -- that
    # stalls every file operation in a tmpfs directory?  tmpfs keeps no negative dentries: each failing lookup takes the directory's rwsem
    # shared and allocates a dentry, and whoever has to MODIFY the directory meanwhile (the daemon's link() for the ending signal, its
    # unlink of pred_start.sig, the encoder's own remove / create) queues behind a stream of readers.  A DIAGNOSTIC client that sleeps
    # between two attempts (not what the unchanged encoder does) takes that stream away:
    if os.path.isdir("/dev/shm"):
        for poll in ("0", "20", "100"):
            run("/dev/shm", 1920, 1080, frames, 0, (), ("--poll-us", poll))
        run("/dev/shm", 416, 240, frames, 0, (), ("--poll-us", "20"))


if __name__ == "__main__":
    main()

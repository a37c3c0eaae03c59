"""Why plan 3's trunk and FC1 are not overlapped: the experiment and the numbers.

    python scripts/plan3_power.py [seconds per setting]     -> gpurun_out/plan3_overlap.txt (profiles/r06_plan3_overlap.txt)

1. Socket power and shader clock (hwmon, 20 ms samples, steady state) of a C3 step looped under plan 3 and plan 0, and of EACH STAGE of
   plan 3 looped alone (experiments build, ETHCNN_STAGE_ONLY): which stages sit at the 1400 W cap, what a step costs in joules.
2. The overlap itself, with the kernels as they are: TWO contexts on the one GPU, each looping plan-3 steps from its own thread on its own
   streams -- the trunk of one runs beside FC1 / heads of the other (CU by CU: neither kernel leaves room for the other inside a CU,
   255 + 2 x 231 VGPRs, 77 + 96 KB LDS; across CUs they share the chip, its clock and its power budget).  Aggregate CTU/s against one
   context alone.  If the chip were short of issue slots or matrix-pipe time per stage, two interleaved streams would fill the gaps; if
   it is short of WATTS, they cannot."""
import ctypes
import glob
import importlib
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else 3.0
W, H, NF, QP = 3840, 2160, 50, 32
NCTU = 60 * 34 * NF


def hwmon_of_device0():
    hw = [h for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*") if os.path.exists(h + "/power1_average") or os.path.exists(h + "/power1_input")]
    mine = None
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, 0) == 0:
            bdf = buf.value.decode().lower()
            for h in hw:
                if os.path.realpath(os.path.join(h, "..", "..")).lower().endswith(bdf):
                    mine = h
    except Exception:
        pass
    return mine, hw


def rd(p):
    try:
        return int(open(p).read().strip())
    except Exception:
        return None


def worker(args):
    """child process: `contexts` contexts (one thread each) loop C3 steps under `plan` for `seconds`; prints one JSON line"""
    plan, contexts, seconds = int(args[0]), int(args[1]), float(args[2])
    pkg = importlib.import_module("hevc-complexity-reduction_amd")
    rng = np.random.default_rng(3)
    frame = rng.integers(0, 256, size=(H, W), dtype=np.uint8)
    frame[: H // 2] = (frame[: H // 2] // 16 + 100).astype(np.uint8)
    luma = np.stack([np.roll(frame, 131 * k, axis=1) for k in range(NF)])
    ctxs = []
    for _ in range(contexts):
        c = pkg.EthCnn(device=0)
        c.load_synthetic(1, 8.0)
        c.set_thresholds(0.5, 0.5)
        c.set_fc1_plan(plan)
        d_in, d_out = c.alloc(luma.nbytes), c.alloc(NCTU * 21 * 4)
        d_in.upload(luma)
        for _ in range(3):
            c.predict_luma_device(d_in, W, H, NF, QP, d_out)
        c.synchronize()
        ctxs.append((c, d_in, d_out))
    steps = [0] * contexts
    stop = time.perf_counter() + seconds
    def loop(i):
        c, d_in, d_out = ctxs[i]
        while time.perf_counter() < stop:
            for _ in range(8):  # a few steps in flight, then wait: the queue never runs dry, the host never runs far ahead
                c.predict_luma_device(d_in, W, H, NF, QP, d_out)
            c.synchronize()
            steps[i] += 8
    t0 = time.perf_counter()
    th = [threading.Thread(target=loop, args=(i,)) for i in range(contexts)]
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    print(json.dumps({"steps": sum(steps), "seconds": dt, "ms_per_step": dt / max(1, sum(steps)) * 1e3, "ctus_per_s": sum(steps) * NCTU / dt}))


def measure(label, plan, contexts, env_extra=None):
    mine, hw = hwmon_of_device0()
    env = dict(os.environ)
    env.update(env_extra or {})
    p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(plan), str(contexts), str(SECONDS)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    rows = []
    t0 = time.time()
    while p.poll() is None:
        h = mine or max(hw, key=lambda x: rd(x + ("/power1_average" if os.path.exists(x + "/power1_average") else "/power1_input")) or 0)
        pw = rd(h + ("/power1_average" if os.path.exists(h + "/power1_average") else "/power1_input"))
        rows.append((time.time() - t0, (pw or 0) / 1e6, (rd(h + "/freq1_input") or 0) / 1e6))
        time.sleep(0.02)
    out, err = p.communicate()
    if p.returncode != 0:
        return "%-44s FAILED: %s" % (label, err[-300:])
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    # steady state: the samples of the last 60 % of the loop (the loop ends ~0.3 s before the process does)
    busy = [r for r in rows if r[1] > 0.6 * max(x[1] for x in rows)]
    tail = busy[len(busy) * 4 // 10:] or busy
    pw = sorted(r[1] for r in tail)[len(tail) // 2]
    ck = sorted(r[2] for r in tail)[len(tail) // 2]
    return ("%-44s %7.4f ms/step  %6.1f M CTU/s  %6.0f W  %5.0f MHz  -> %6.3f J per step" %
            (label, d["ms_per_step"], d["ctus_per_s"] / 1e6, pw, ck, pw * d["ms_per_step"] * 1e-3)), d, pw, ck


def main():
    exp = os.path.join(ROOT, "hevc-complexity-reduction_amd", "lib_exp", "libethcnn.so")
    mine, hw = hwmon_of_device0()
    cap = rd((mine or hw[0]) + "/power1_cap")
    lines = ["# scripts/plan3_power.py: C3 steps (102,000 CTUs) looped for %.0f s per setting; socket power / shader clock = median of the hwmon samples (20 ms apart)" % SECONDS,
             "# of the loop's steady state; power cap %.0f W" % ((cap or 0) / 1e6), ""]
    res = {}
    for label, plan, nctx, extra in (("plan 0 (exact), one context", 0, 1, None),
                                     ("plan 3, one context", 3, 1, None),
                                     ("plan 3, TWO contexts interleaved", 3, 2, None),
                                     ("plan 3, FOUR contexts interleaved", 3, 4, None),
                                     ("plan 0 (exact), TWO contexts interleaved", 0, 2, None),
                                     ("plan 3, trunk stage alone, looped", 3, 1, {"ETHCNN_LIB": exp, "ETHCNN_STAGE_ONLY": "1"}),
                                     ("plan 3, FC1 stage alone, looped", 3, 1, {"ETHCNN_LIB": exp, "ETHCNN_STAGE_ONLY": "2"}),
                                     ("plan 3, heads + gate alone, looped", 3, 1, {"ETHCNN_LIB": exp, "ETHCNN_STAGE_ONLY": "3"}),
                                     ("plan 0, trunk (+ CTU load) alone, looped", 0, 1, {"ETHCNN_LIB": exp, "ETHCNN_STAGE_ONLY": "1"}),
                                     ("plan 0, FC1 stage alone, looped", 0, 1, {"ETHCNN_LIB": exp, "ETHCNN_STAGE_ONLY": "2"})):
        r = measure(label, plan, nctx, extra)
        if isinstance(r, str):
            lines.append(r)
        else:
            lines.append(r[0])
            res[label] = r[1:]
        print(lines[-1], flush=True)
    text = "\n".join(lines)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "plan3_overlap.txt"), "w").write(text + "\n")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(sys.argv[2:])
    else:
        main()

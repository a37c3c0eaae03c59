#!/usr/bin/env python
"""Reads a rocprofv3 --kernel-trace csv of bench.py and reports how the stages of consecutive passes overlapped:
busy time per kernel family, time with >= 2 kernels resident, idle gaps, and the step period."""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    fam = "tile" if "k0_tile" in n else "trunk" if "k1_trunk" in n else "fc1" if "k_fc1" in n else "heads" if "k_heads" in n else "gate" if "k5_gate" in n else None
    if fam: rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), fam))
rows.sort()
rows = rows[len(rows) // 3:]  # skip ramp / warm-up
t0, t1 = rows[0][0], max(r[1] for r in rows)
ev = []
for s, e, fam in rows: ev += [(s, 1, fam), (e, -1, fam)]
ev.sort()
active = collections.Counter(); last = t0; busy1 = busy2 = idle = 0
pair = collections.Counter()
for t, d, fam in ev:
    k = sum(active.values()); dt = t - last
    if k == 0: idle += dt
    elif k == 1: busy1 += dt
    else:
        busy2 += dt
        pair[tuple(sorted(a for a in active if active[a] > 0))] += dt
    active[fam] += d; last = t
dur = collections.Counter(); cnt = collections.Counter()
for s, e, fam in rows: dur[fam] += e - s; cnt[fam] += 1
nfc1 = sum(1 for r in rows if r[2] == "fc1") / 2.0  # main + remainder dispatch per step
print("window %.3f ms, ~%.1f steps, period %.3f ms/step" % ((t1 - t0) / 1e6, nfc1, (t1 - t0) / 1e6 / max(nfc1, 1)))
print("exactly one kernel resident %.1f %%   two or more %.1f %%   none (gaps) %.1f %%" % (100.0 * busy1 / (t1 - t0), 100.0 * busy2 / (t1 - t0), 100.0 * idle / (t1 - t0)))
print("average kernel duration (us):", {k: round(dur[k] / cnt[k] / 1e3, 1) for k in dur})
print("concurrent pairs (ms):", {"+".join(k): round(v / 1e6, 3) for k, v in pair.most_common(6)})

"""Where a step's time goes between its kernels (C2 runs at 0.74 of the fp32-MFMA ceiling against C3's 0.82).

    python scripts/trace_gaps.py <..._kernel_trace.csv> [label]

From a rocprofv3 --kernel-trace CSV of `bench.py --workload <wl> --no-other-configs --no-fast-plan`: the main compute stream is the one
that runs the FC1 kernel; a STEP is the span from one trunk launch to the next on that stream.  Over the steady steps (the longest run of
equal-period steps = the timed region): kernel time per stage, idle time on the main stream per boundary (trunk -> FC1, FC1 -> heads,
heads -> gate, gate -> next trunk), the step period, and how much of the side stream's CTU-load kernel falls inside FC1."""
import collections
import csv
import statistics
import sys


def short(name):
    return name.replace("void ", "").replace("ethcnn::", "").split("(")[0].split("<")[0]


def main(path, label=""):
    rows = []
    for r in csv.DictReader(open(path)):
        if "ethcnn" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Stream_Id") or r.get("Queue_Id")))
    rows.sort()
    fc1_streams = collections.Counter(s for _, _, k, s in rows if k.startswith("k_fc1"))
    main_s = fc1_streams.most_common(1)[0][0]
    ms = [r for r in rows if r[3] == main_s]
    side = [r for r in rows if r[3] != main_s and r[2].startswith("k0_tile")]
    # steps: from a trunk launch to the next one on the main stream
    idx = [i for i, r in enumerate(ms) if r[2].startswith("k1_trunk")]
    steps = []
    for a, b in zip(idx, idx[1:]):
        ks = ms[a:b]
        names = [k[2] for k in ks]
        if len(ks) < 4 or not any(n.startswith("k_fc1") for n in names):
            continue
        period = ms[b][0] - ks[0][0]
        steps.append((period, ks))
    if not steps:
        raise SystemExit("no steps found")
    med = statistics.median(p for p, _ in steps)
    steady = [(p, ks) for p, ks in steps if abs(p - med) < 0.15 * med and len(ks) == statistics.mode(len(k) for _, k in steps)]
    stage_t = collections.defaultdict(list)
    gaps = collections.defaultdict(list)
    for p, ks in steady:
        for k in ks:
            stage_t[k[2]].append(k[1] - k[0])
        for a, b in zip(ks, ks[1:]):
            gaps["%s -> %s" % (a[2], b[2])].append(b[0] - a[1])
        gaps["%s -> next k1_trunk" % ks[-1][2]].append(ks[0][0] + p - ks[-1][1])
    n = len(steady)
    print("%s: %d steady steps of %d kernels on the main stream (of %d steps in the trace); step period %.1f us (median)" % (label or path, n, len(steady[0][1]), len(steps), med / 1e3))
    tot_k = 0.0
    for k, v in stage_t.items():
        m = sum(v) / n / 1e3
        tot_k += m
        print("   kernel  %-22s %8.1f us per step (%d launch%s)" % (k, m, round(len(v) / n), "" if round(len(v) / n) == 1 else "es"))
    tot_g = 0.0
    for k, v in gaps.items():
        m = sum(v) / n / 1e3
        tot_g += m
        print("   idle    %-44s %6.1f us per step (min %.1f, max %.1f)" % (k, m, min(v) / 1e3, max(v) / 1e3))
    print("   = kernels %.1f us + idle %.1f us = %.1f us; idle = %.1f %% of the period" % (tot_k, tot_g, tot_k + tot_g, 100.0 * tot_g / (tot_k + tot_g)))
    if side:
        # overlap of the side-stream CTU-load kernels with main-stream kernels, by stage
        ov = collections.defaultdict(float)
        t0, t1 = steady[0][1][0][0], steady[-1][1][-1][1]
        sk = [s for s in side if t0 <= s[0] <= t1]
        for s in sk:
            for p, ks in steady:
                for k in ks:
                    o = min(s[1], k[1]) - max(s[0], k[0])
                    if o > 0:
                        ov[k[2]] += o
        tot = sum(s[1] - s[0] for s in sk) or 1
        print("   side stream: %d CTU-load launches, %.1f us each; of their time %s ran beside main-stream kernels" %
              (len(sk), tot / max(1, len(sk)) / 1e3, ", ".join("%.0f %% %s" % (100.0 * v / tot, k) for k, v in sorted(ov.items(), key=lambda kv: -kv[1]))))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")

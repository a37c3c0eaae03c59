"""HBM-side traffic per kernel AND grid size from rocprofv3 PMC passes -> profiles/step_traffic.json + a tracked CSV.

VERDICT r05 item 1: round 5's `roofline.traffic` averaged FETCH_SIZE / WRITE_SIZE over every k_fc1* dispatch of a bench run that also
visited the other configs (three grid sizes in one average).  Here every row is ONE (kernel, grid size) of ONE workload:

    python scripts/pmc_traffic.py collect <workload> <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass>

reads the two counter_collection.csv files of `bench.py --workload <wl> --no-other-configs ...` runs (scripts/gpu_step_traffic.sh),
writes gpurun_out/traffic_by_kernel_grid_<wl>.csv (one row per kernel x grid: dispatches, FETCH_SIZE / WRITE_SIZE averages as
reported in KB, bytes = FETCH x 2 x 1024 + WRITE x 1024 -- the gfx950 correction of MI355X_MICROARCH.md) and merges the per-step sums
of every plan into gpurun_out/step_traffic.json:  <wl>, <wl>_plan2, <wl>_plan3 -> per_kernel {"kernel @grid": {...}}, bytes_per_step,
fc1_bytes_per_launch (= the FC1 rows of that step: what bench.py prints as roofline.traffic AND finds again under
hbm.per_kernel_bytes -- the same numbers by construction, asserted in tests/test_gpu_bench_contract.py).

A step of plan P = every (kernel, grid) row of the kernels of P; how often a row runs per step = its dispatch count / the dispatch
count of the plan's anchor kernel (the trunk form only that plan launches), which must come out as a whole number."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CTUS = {"c2": 25500, "c3": 102000, "c4": 54 * 3927}
# kernels of a step, by plan: (name prefix, shared with other plans?)  The first entry is the plan's anchor.
PLANS = {
    "": ["k1_trunk<false, 0>", "k0_tile_slab", "k_fc1_bulk", "k_fc1_p3", "k_fc1_regs", "k_heads", "k5_gate"],
    "_plan2": ["k1_trunk<false, 2>", "k0_tile_slab", "k_fc1_fast", "k_heads", "k5_gate"],
    "_plan3": ["k1_trunk_f16_foldall", "k_fc1_fast", "k_heads_f16", "k5_gate"],
}
SHARED = ("k0_tile_slab", "k_heads", "k5_gate", "k_fc1_fast")  # launched by more than one plan: once per step in each


def match(pat, k):
    """`pat` names kernel k: the same name, the same template, or (pattern with its template arguments) a prefix"""
    return k == pat or k.startswith(pat + "<") or ("<" in pat and k.startswith(pat))


def short(name):
    return name.replace("void ", "").replace("ethcnn::", "").split("(")[0]


def read_pass(d, counter):
    """{(kernel, grid): [sum, dispatches]} of one counter"""
    agg = collections.defaultdict(lambda: [0.0, 0])
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no counter_collection.csv under %s" % d)
    for fn in files:
        for r in csv.DictReader(open(fn)):
            if "ethcnn" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                k = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
                agg[k][0] += float(r["Counter_Value"])
                agg[k][1] += 1
    return agg


def collect(wl, dir_fetch, dir_write):
    import bench
    fe, wr = read_pass(dir_fetch, "FETCH_SIZE"), read_pass(dir_write, "WRITE_SIZE")
    rows = {}
    for k in sorted(set(fe) | set(wr)):
        f_kb = fe[k][0] / fe[k][1] if k in fe and fe[k][1] else 0.0
        w_kb = wr[k][0] / wr[k][1] if k in wr and wr[k][1] else 0.0
        rows[k] = {"dispatches": max(fe.get(k, [0, 0])[1], wr.get(k, [0, 0])[1]), "fetch_kb": f_kb, "write_kb": w_kb,
                   "fetch_bytes": int(f_kb * 1024 * 2), "write_bytes": int(w_kb * 1024)}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    csv_path = os.path.join(ROOT, "gpurun_out", "traffic_by_kernel_grid_%s.csv" % wl)
    with open(csv_path, "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of `python bench.py --workload %s "
                "--no-cpu-baseline --no-host-scopes --no-other-configs --fast-plans 2,3 --steps 3 --warmup 1` (scripts/gpu_step_traffic.sh).\n"
                "# One row per kernel AND grid size; *_kb = per-dispatch average as reported; bytes = FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024 "
                "(gfx950: FETCH_SIZE under-counts wide reads by 2, MI355X_MICROARCH.md).\n" % wl)
        f.write("workload,kernel,grid_size,dispatches,fetch_size_kb_avg,write_size_kb_avg,bytes_per_dispatch\n")
        for (k, g), r in rows.items():
            f.write('%s,"%s",%d,%d,%.3f,%.3f,%d\n' % (wl, k, g, r["dispatches"], r["fetch_kb"], r["write_kb"], r["fetch_bytes"] + r["write_bytes"]))
    out_path = os.path.join(ROOT, "gpurun_out", "step_traffic.json")
    try:
        out = json.load(open(out_path))
    except Exception:
        out = {}
    stamp = bench.kernel_source_stamp()
    if out.get("kernel_source_stamp") != stamp:
        out = {}
    out["kernel_source_stamp"] = stamp
    out["source"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (kernel trace only) of `python bench.py --workload <wl> --no-cpu-baseline "
                     "--no-host-scopes --no-other-configs --fast-plans 2,3 --steps 3 --warmup 1` (scripts/gpu_step_traffic.sh -> scripts/pmc_traffic.py): "
                     "per-dispatch averages per kernel AND grid size (profiles/*traffic_by_kernel_grid_<wl>.csv), summed over the dispatches of one "
                     "step of the plan; FETCH_SIZE x2 (gfx950 correction)")
    n = CTUS[wl]
    for suffix, pats in PLANS.items():
        anchor = [(k, r) for k, r in rows.items() if match(pats[0], k[0])]
        if not anchor:
            print("%s%s: no dispatch of %s -- plan not visited" % (wl, suffix, pats[0]))
            continue
        steps = max(r["dispatches"] for _, r in anchor)  # the anchor runs once per pass (c2 / c3: one pass per step); its other grid sizes
        # are the accuracy guard's calibration passes
        per, fc1 = {}, 0
        for (k, g), r in rows.items():
            pat = next((p for p in pats if match(p, k)), None)
            if pat is None:
                continue
            # a kernel more than one plan launches runs once per step; of its grid sizes the step uses the one the timed region uses
            # (k0_tile_slab: the persistent side-stream form; the other grid belongs to the three pipeline-off profiling steps)
            if pat in SHARED and any(match(pat, k2) and r2["dispatches"] > r["dispatches"] for (k2, _), r2 in rows.items()):
                continue
            times = 1.0 if pat in SHARED else r["dispatches"] / float(steps)
            if times < 0.5:
                # not a kernel of the step: the 640-CTU calibration passes of the plans' accuracy guard (once per weight load), first-frame checks
                print("  (not part of a step: %s @%d, %d dispatches in %d steps)" % (k, g, r["dispatches"], steps))
                continue
            if abs(times - round(times)) < 0.1:
                times = float(round(times))  # (a calibration pass may share a grid size with a kernel of the step: 26 dispatches in 24 steps)
            else:
                print("  note: %s @%d runs %.3f times per step of %s%s" % (k, g, times, wl, suffix))
            b = int((r["fetch_bytes"] + r["write_bytes"]) * times)
            per["%s @%d" % (k, g)] = {"grid_size": g, "per_step": times, "fetch_bytes": int(r["fetch_bytes"] * times),
                                       "write_bytes": int(r["write_bytes"] * times), "bytes": b}
            if k.startswith("k_fc1"):
                fc1 += b
        alg_fc1 = (n * 2688 * 4 + 2688 * 448 * 4 + n * 448 * 4) if suffix == "" else (n * 2688 * 2 * 2 + 2688 * 448 * 2 * 2 + n * 448 * 4)
        out[wl + suffix] = {"bytes_per_step": sum(v["bytes"] for v in per.values()), "per_kernel": per, "algorithmic_bytes_per_step": n * 4180,
                            "fc1_bytes_per_launch": fc1, "fc1_algorithmic_bytes_per_launch": alg_fc1, "steps_seen": steps}
        print("%s%s: %.3f GB per step, FC1 %.3f GB = %.2fx algorithmic" % (wl, suffix, out[wl + suffix]["bytes_per_step"] / 1e9, fc1 / 1e9, fc1 / alg_fc1),
              {k: "%.3f" % (v["bytes"] / 1e9) for k, v in per.items()})
    json.dump(out, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    if len(sys.argv) == 5 and sys.argv[1] == "collect":
        collect(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        raise SystemExit(__doc__)

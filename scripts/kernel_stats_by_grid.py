"""rocprofv3 kernel trace -> one row per kernel AND grid size (`--stats` lumps the grid sizes of a kernel, so
"k_fc1_bulk: 143 calls, 1.707 ms" mixed c3's, c2's and c4's launches).

    python scripts/kernel_stats_by_grid.py <..._kernel_trace.csv> <out.csv> ["comment for the header"]

Columns: kernel, grid (x*y*z work-items), workgroup, calls, total_ms, avg_us, min_us, max_us, stddev_us, pct of the listed time."""
import collections
import csv
import math
import sys


def short(name):
    return name.replace("void ", "").replace("ethcnn::", "").split("(")[0]


def main(trace, out, comment=""):
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        if r.get("Kind", "KERNEL_DISPATCH") != "KERNEL_DISPATCH":
            continue
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        rows[(short(r["Kernel_Name"]), grid, wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    total = sum(sum(v) for v in rows.values()) or 1.0
    with open(out, "w") as f:
        if comment:
            f.write("# %s\n" % comment)
        f.write("# one row per kernel AND grid size, from the rocprofv3 --kernel-trace CSV of the command above (scripts/kernel_stats_by_grid.py)\n")
        f.write("kernel,grid_size,workgroup_size,calls,total_ms,avg_us,min_us,max_us,stddev_us,pct\n")
        for (k, g, wg), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
            n, s = len(v), sum(v)
            mean = s / n
            sd = math.sqrt(sum((x - mean) ** 2 for x in v) / n)
            f.write('"%s",%d,%d,%d,%.3f,%.3f,%.3f,%.3f,%.3f,%.2f\n' % (k, g, wg, n, s / 1e3, mean, min(v), max(v), sd, 100.0 * s / total))


if __name__ == "__main__":
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")

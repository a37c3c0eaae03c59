"""socket power / shader clock (hwmon sysfs, 20 ms samples) while bench.py runs its plan-0 region and then its plan-1 region:
is FC1 plan 1 held back by the power cap?  python scripts/power_probe.py [steps]"""
import glob, os, subprocess, sys, time
steps = sys.argv[1] if len(sys.argv) > 1 else "2500"
hw = [h for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*") if os.path.exists(h + "/power1_average") or os.path.exists(h + "/power1_input")]
print("hwmon:", hw)
def rd(p):
    try:
        return int(open(p).read().strip())
    except Exception:
        return None
def pwf(h):
    return h + ("/power1_average" if os.path.exists(h + "/power1_average") else "/power1_input")
# the job's GPU among the node's: by PCI address (HIP sees one device, sysfs all eight)
import ctypes
mine = None
try:
    hip = ctypes.CDLL("libamdhip64.so")
    buf = ctypes.create_string_buffer(64)
    if hip.hipDeviceGetPCIBusId(buf, 64, 0) == 0:
        bdf = buf.value.decode().lower()
        for h in hw:
            if os.path.realpath(os.path.join(h, "..", "..")).lower().endswith(bdf):
                mine = h
except Exception as exc:
    print("PCI lookup failed:", exc)
cap = rd((mine or hw[0]) + "/power1_cap")
print("cap W:", cap and cap / 1e6, " job's GPU:", mine or "unknown (the busiest of %d is taken)" % len(hw))
p = subprocess.Popen([sys.executable, "bench.py", "--no-cpu-baseline", "--no-host-scopes", "--steps", steps], stdout=open("gpurun_out/power_bench.json", "w"), stderr=subprocess.DEVNULL)
t0 = time.time()
rows = []
while p.poll() is None:
    best = mine or max(hw, key=lambda h: rd(pwf(h)) or 0)
    rows.append((time.time() - t0, rd(pwf(best)), rd(best + "/freq1_input"), best.split("/")[4]))
    time.sleep(0.02)
print("samples", len(rows), "(20 ms apart; bench.py runs its exact region, then plan 1, then plan 2)")
for i in range(0, len(rows), max(1, len(rows) // 60)):
    t, w, f, name = rows[i]
    print("t %6.2f s  %s power %7.1f W  sclk %s MHz" % (t, name, (w or 0) / 1e6, (f or 0) / 1e6))

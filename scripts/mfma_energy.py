"""Joules per TFLOP of gfx950's matrix instructions by shape / type on random operands (round 6; plan 3 is bound by the socket's power cap).

    python scripts/mfma_energy.py [seconds per kind]   -> gpurun_out/mfma_energy.txt (profiles/r06_mfma_energy.txt)

scripts/ubench/mfma_energy.hip loops one instruction (3 waves per SIMD, 4 independent accumulators, operands that change every iteration) while
this script samples the socket's power and shader clock (hwmon, 20 ms); steady state = the busy samples' median."""
import glob
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from plan3_power import hwmon_of_device0, rd  # noqa: E402

SEC = sys.argv[1] if len(sys.argv) > 1 else "4"
KINDS = [(0, "v_mfma_f32_16x16x4_f32   (exact plan)"), (1, "v_mfma_f32_16x16x16_f16"), (2, "v_mfma_f32_16x16x32_f16  (plan 3 trunk / heads)"),
         (3, "v_mfma_f32_32x32x8_f16"), (4, "v_mfma_f32_32x32x16_f16  (plan 2 / 3 FC1)"), (5, "v_mfma_f32_32x32x16_bf16"),
         (6, "v_mfma_scale_f32_32x32x64_f8f6f4 (fp8)"), (7, "v_mfma_i32_32x32x32_i8")]


def main():
    exe = os.path.join(ROOT, "scripts", "ubench", "mfma_energy")
    if not os.path.exists(exe):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-o", exe, exe + ".hip"])
    mine, hw = hwmon_of_device0()
    h = mine or hw[0]
    pf = h + ("/power1_average" if os.path.exists(h + "/power1_average") else "/power1_input")
    idle = (rd(pf) or 0) / 1e6
    lines = ["# scripts/mfma_energy.py: one matrix instruction looped for %s s (3 waves per SIMD, 4 accumulators, operands changing every iteration); socket power / shader clock =" % SEC,
             "# median of the busy hwmon samples; idle socket %.0f W; cap %.0f W.  J per TFLOP = W / (TFLOP/s); 'dynamic' subtracts the idle socket." % (idle, (rd(h + "/power1_cap") or 0) / 1e6),
             "%-52s %10s %8s %8s %12s %14s" % ("instruction", "TFLOP/s", "W", "MHz", "J per TFLOP", "dynamic J/TFLOP")]
    for kind, name in KINDS:
        p = subprocess.Popen([exe, str(kind), SEC], stdout=subprocess.PIPE, text=True)
        rows = []
        while p.poll() is None:
            rows.append(((rd(pf) or 0) / 1e6, (rd(h + "/freq1_input") or 0) / 1e6))
            time.sleep(0.02)
        tf = float(p.stdout.read().strip() or 0)
        busy = [r for r in rows if r[0] > 0.6 * max(x[0] for x in rows)]
        tail = busy[len(busy) // 3:] or busy
        w = sorted(r[0] for r in tail)[len(tail) // 2]
        mhz = sorted(r[1] for r in tail)[len(tail) // 2]
        lines.append("%-52s %10.1f %8.0f %8.0f %12.3f %14.3f" % (name, tf, w, mhz, w / tf if tf else 0, (w - idle) / tf if tf else 0))
        print(lines[-1], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "mfma_energy.txt"), "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()

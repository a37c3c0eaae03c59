import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/v*.json")):
    try:
        d = json.load(open(f))
        box = d["roofline"].get("box_mfma_tflops")
        print("%s: value %.2fM CTU/s  fc1 %.3f ms (%.1f TF, frac %.3f%s)  stages %s parity %s" % (
            f.split("/")[-1], d["value"] / 1e6, d["roofline"]["avg_launch_ms"], d["roofline"]["achieved"], d["roofline"]["frac"],
            "; box %.1f TF -> %.3f" % (box, d["roofline"]["achieved"] / box) if box else "",
            {k: round(x, 3) for k, x in d["stages_ms_per_step"].items()}, d.get("parity_first_frame_bit_exact")))
    except Exception as e:
        print(f, "failed:", e)

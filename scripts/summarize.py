import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/v*.json")):
    try:
        d = json.load(open(f))
        box = d["roofline"].get("box_mfma_tflops")
        print("%s: value %.2fM CTU/s  fc1 %.3f ms (%.1f TF, frac %.3f%s)  stages %s parity %s" % (
            f.split("/")[-1], d["value"] / 1e6, d["roofline"]["avg_launch_ms"], d["roofline"]["achieved"], d["roofline"]["frac"],
            "; box %.1f TF -> %.3f" % (box, d["roofline"]["achieved"] / box) if box else "",
            {k: round(x, 3) for k, x in d["stages_ms_per_step"].items()}, d.get("parity_first_frame_bit_exact")))
        for key in ("fast_plan", "fast_plan_fp16x2", "fast_plan_fp16x2_trunk"):
          fp = d.get(key)
          if fp:
            print("    " + key + ": value %.2fM CTU/s  %.3f ms/step  fc1 %.3f ms (frac of bf16 peak %.3f)  stages %s  max|d| %s flips %s" % (
                fp["value"] / 1e6, fp["ms_per_step"], fp["roofline"]["avg_launch_ms"], fp["roofline"]["frac"],
                {k: round(x, 3) for k, x in fp["stages_ms_per_step"].items()}, fp["max_abs_vs_exact"], fp["flips_vs_exact"]))
    except Exception as e:
        print(f, "failed:", e)

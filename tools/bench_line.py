"""prints the headline numbers of a bench.py JSON line read from stdin (diagnostics)"""
import json
import sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d["roofline"]
print(sys.argv[1] if len(sys.argv) > 1 else "", "fps", d["value"], "ms", d["ms_per_step"], "kernel_only", d["mlp_msamples_per_s_kernel_only"],
      "launch_ms", r["avg_launch_ms"], "frac", r["frac"], "frame_frac", r.get("frame_frac"), "samples", d["samples_per_frame"],
      "iters", d["iterations_per_frame"], "psnr", d.get("psnr_db"))

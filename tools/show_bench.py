"""Prints the key figures of a bench.py JSON line: python tools/show_bench.py <file>"""
import json
import sys

d = json.load(open(sys.argv[1]))
print("value %.0f images/s, incl. host io %s, ms/step %.1f" % (d["value"], d.get("value_incl_host_io"), d["ms_per_step"]))
if "roofline" in d:
    r = d["roofline"]
    print("roofline %s frac %.3f, %.1f us per search launch, mfma frac %.3f" % (r["bound"], r["frac"], r["avg_launch_us"], r["mfma_frac"]))
for k, c in d.get("configs", {}).items():
    keys = ("images_per_s", "images_per_s_incl_host_io", "us_per_lockstep_iteration", "us_per_iteration", "iterations", "seconds")
    print(k, {q: (round(c[q], 2) if isinstance(c[q], float) else c[q]) for q in keys if c.get(q) is not None},
          ("roofline %s %.3f, search %.1f us" % (c["roofline"]["bound"], c["roofline"]["frac"], c["roofline"]["avg_launch_us"]))
          if c.get("roofline") else "")
    if "split_rccl" in c:
        s = c["split_rccl"]
        print("   split:", {q: s.get(q) for q in ("world", "us_per_iteration", "single_gpu_us_per_iteration", "rccl_ranks_seen", "error")})
    if "next_stage_filter_noise" in c:
        print("   filter_noise:", {q: round(v, 1) for q, v in c["next_stage_filter_noise"].items() if isinstance(v, float)})

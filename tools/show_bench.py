#!/usr/bin/env python3
"""Prints the figures of one or more saved bench lines side by side (tuning aid).
Usage: python tools/show_bench.py gpurun_out/<tag>/bench.json ..."""
import json
import sys

for path in sys.argv[1:]:
    d = json.load(open(path))
    print(path)
    print("  headline %8.1f GB/s  %.4f ms  whole %.4f  dominant %.4f" % (d["value"], d["ms_per_step"], d["whole_batch_hbm_frac"], d["roofline"]["frac"]))
    for k, v in d.get("other_workloads", {}).items():
        ks = " ".join("%s x%d %.1fus" % (x["kernel"], x["launches"], 1000 * x["ms_per_step"]) for x in v["kernels"])
        print("  %-11s %8.1f GB/s  %.4f ms  whole %.4f  last %.4f  | %s" % (k, v["value"], v["ms_per_step"], v["whole_batch_hbm_frac"], v["roofline"]["frac"], ks))

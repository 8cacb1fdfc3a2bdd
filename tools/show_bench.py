#!/usr/bin/env python3
"""Prints the essentials of a bench.py JSON line: python tools/show_bench.py <file>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%.1f Mpaths/s  %.1f ms/step  %s" % (d["value"], d["ms_per_step"], d["config"]["workload"]))
w = d["work_per_pixel_sample"]
print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in w.items()})
for k, v in d["kernels"].items():
    b = v["algorithmic_bytes_per_pixel_sample"]
    print("  %-16s %8.3f ms x %5d   %s B/ps" % (k, v["ms_per_launch"], v["launches"], "-" if b is None else round(b, 1)))
print(d["roofline"])
if "cpu_baseline" in d:
    print({k: v for k, v in d["cpu_baseline"].items() if k not in ("sample", "reference_scan")})
for k, v in d.get("other_configs", {}).items():
    print(k, v.get("value"), v.get("error"))

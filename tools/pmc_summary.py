#!/usr/bin/env python3
"""Summarises the counter CSVs written by tools/pmc_passes.sh: per kernel, the sum of every counter
over its dispatches, plus the number of dispatches. Usage: pmc_summary.py <dir> [out.json]"""
import csv, glob, json, os, sys, collections
d = sys.argv[1]
out = collections.defaultdict(lambda: collections.defaultdict(float))
ndisp = collections.defaultdict(set)
for f in glob.glob(os.path.join(d, "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ub::", "")
        if "anonymous" in k or "rocprim" in k or "rocclr" in k or "at::" in k:
            continue
        out[k][r["Counter_Name"]] += float(r["Counter_Value"])
        ndisp[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
res = {}
for k, cs in out.items():
    res[k] = {c: v for c, v in sorted(cs.items())}
    res[k]["dispatches"] = max(len(ndisp[(k, c)]) for c in cs)
json.dump(res, open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout, indent=1, sort_keys=True)

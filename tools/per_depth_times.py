#!/usr/bin/env python3
"""Per-bounce durations of the wavefront kernels from a rocprofv3 --kernel-trace CSV:
python tools/per_depth_times.py <dir with *_kernel_trace.csv> [launches of trace / shade per batch, default 6]"""
import csv, glob, sys, collections
d = sys.argv[1]
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 6
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = collections.defaultdict(lambda: collections.defaultdict(list))
count = collections.defaultdict(lambda: collections.defaultdict(int))
for r in rows:
    n = r["Kernel_Name"]
    for k in ("k_wf_trace", "k_wf_shade"):
        if k + "<" in n and "<false" in n.replace("ILb0", "<false"):
            s = r.get("Stream_Id", "0")
            i = count[k][s] % depth
            count[k][s] += 1
            per[k][i].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k in per:
    print(k, " ".join("d%d %.3f ms (n=%d)" % (i, sum(v) / len(v), len(v)) for i, v in sorted(per[k].items())))

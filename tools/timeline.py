#!/usr/bin/env python3
"""Per-kernel timeline of the last complete batch in a rocprofv3 --kernel-trace csv (one stream):
  python tools/timeline.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'k_wf_gen<false>' in n]
start, end = idx[-2], idx[-1]
t0 = int(rows[start]['Start_Timestamp'])
prev_end = t0
tot = {}
for r in rows[start:end]:
    n = r['Kernel_Name'].split('(')[0].replace('void ub::', '').replace('ub::', '')
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if 'fillBuffer' not in n:
        print("%-32s start %8.3f dur %8.3f gap %7.3f" % (n[:32], (s - t0) / 1e6, (e - s) / 1e6, (s - prev_end) / 1e6))
    tot[n.split('<')[0]] = tot.get(n.split('<')[0], 0) + (e - s) / 1e6
    prev_end = e
print("batch total %.3f ms" % ((prev_end - t0) / 1e6), {k: round(v, 2) for k, v in tot.items() if 'fill' not in k})

#!/bin/bash
# Sweep of lane count, batch size and wavefront depth on the GPU box (C2): bash tools/sweep_lanes.sh
set -e
for cfg in "default:" "s1:--streams 1" "s3:--streams 3" "s4:--streams 4" "pif8_s2:--passes-in-flight 8 --streams 2" \
           "pif8_s4:--passes-in-flight 8 --streams 4" "pif32_s2:--passes-in-flight 32 --streams 2 --passes-per-step 32 --steps 8" \
           "d6:--wavefront-depth 6" "d10:--wavefront-depth 10" "d12:--wavefront-depth 12"; do
  n=${cfg%%:*}; a=${cfg#*:}
  timeout -k 10 200 python3 bench.py --no-cpu-baseline $a > gpurun_out/sl_$n.log 2> gpurun_out/sl_$n.err
  python3 - "$n" <<'PY'
import json,sys
n=sys.argv[1]
d=json.loads(open(f"gpurun_out/sl_{n}.log").read().strip().splitlines()[-1])
print("%-10s %.1f Mpaths/s"%(n, d["value"]), {k: round(v["ms_per_launch"],3) for k,v in d["kernels"].items()})
PY
done

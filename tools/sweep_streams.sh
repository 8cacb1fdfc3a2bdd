#!/bin/bash
# Sweep of scheduling knobs on the GPU box (C2): bash tools/sweep_streams.sh
set -e
for cfg in "s2d8p4:--streams 2 --wavefront-depth 8" "s2d8p8:--streams 2 --wavefront-depth 8 --passes-in-flight 8" "s3d8p8:--streams 3 --wavefront-depth 8 --passes-in-flight 8" "s2d8p16:--streams 2 --wavefront-depth 8 --passes-in-flight 16" "s3d8p12:--streams 3 --wavefront-depth 8 --passes-in-flight 12" "s2d6p16:--streams 2 --passes-in-flight 16" "s1d8p16:--streams 1 --wavefront-depth 8 --passes-in-flight 16"; do
  n=${cfg%%:*}; a=${cfg#*:}
  timeout -k 10 200 python3 bench.py --no-cpu-baseline $a > gpurun_out/sw_$n.log 2> gpurun_out/sw_$n.err
  python3 - "$n" <<'PY'
import json,sys
n=sys.argv[1]
d=json.loads(open(f"gpurun_out/sw_{n}.log").read().strip().splitlines()[-1])
print("%-8s %.1f Mpaths/s"%(n, d["value"]))
PY
done

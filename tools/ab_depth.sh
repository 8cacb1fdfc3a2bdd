#!/bin/bash
# A/B of the number of wavefront bounces on C2, three runs each, alternating.
set -u
mkdir -p gpurun_out/ab_depth
for r in 1 2 3; do
  for d in 8 10; do
    timeout -k 10 120 python3 bench.py --no-cpu-baseline --wavefront-depth $d > gpurun_out/ab_depth/d${d}_$r.json 2> gpurun_out/ab_depth/d${d}_$r.err || exit 1
    python3 -c "import json; d=json.loads(open('gpurun_out/ab_depth/d${d}_$r.json').read().strip().splitlines()[-1]); print('depth $d run $r', round(d['value'],1))"
  done
done

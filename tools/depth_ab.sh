#!/bin/bash
# Wavefront depth sweep on one workload: bash tools/depth_ab.sh "<depths>" [bench args]
DEPTHS=$1; shift
for d in $DEPTHS; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-other-configs --wavefront-depth $d "$@" > gpurun_out/depth.log 2> gpurun_out/depth.err || { tail -3 gpurun_out/depth.err; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/depth.log').read().strip().splitlines()[-1]);print('depth $d', round(d['value'],1), {k.replace('k_',''): round(v['ms_per_launch'],2) for k,v in d['kernels'].items() if 'tail' in k or 'trace' in k})"
done

#!/bin/bash
# Batch-size / lane / depth sweep on the GPU box: bash tools/batch_sweep.sh "<label> <bench args>" ...
# prints Mpaths/s, launches of k_wf_gen (= batches) and the samples per pixel of each run
for spec in "$@"; do
  set -- $spec
  l=$1; shift
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-other-configs "$@" > gpurun_out/sw.json 2> gpurun_out/sw.err || { echo "$l failed"; tail -3 gpurun_out/sw.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/sw.json').read().strip().splitlines()[-1]); print('$l', round(d['value'],1), d['kernels']['k_wf_gen']['launches'], d['config']['spp_total'])"
done

#!/bin/bash
# Scheduling knobs once more on the final kernels: lanes, wavefront depth, batch size (C2 512x512).
set -u
mkdir -p gpurun_out/sweep_final
run() { tag=$1; shift; timeout -k 10 120 python3 bench.py --no-cpu-baseline "$@" > gpurun_out/sweep_final/$tag.json 2> gpurun_out/sweep_final/$tag.err || return 1
  python3 -c "import json,sys; d=json.loads(open('gpurun_out/sweep_final/$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['value'],1))"; }
run base && run s3 --streams 3 && run s4 --streams 4 && run d6 --wavefront-depth 6 && run d10 --wavefront-depth 10 && run d12 --wavefront-depth 12 \
 && run p32 --passes-per-step 32 --passes-in-flight 32 --steps 8 && run p32s3 --passes-per-step 32 --passes-in-flight 32 --steps 8 --streams 3 && run p8 --passes-per-step 16 --passes-in-flight 8

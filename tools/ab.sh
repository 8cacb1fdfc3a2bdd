#!/bin/bash
# A/B of two builds of the library on the GPU box: bash tools/ab.sh <tagA> <tagB> [bench args]
# (tag "" = lib/libubdpt.so, tag "_x" = lib/libubdpt_x.so); prints Mpaths/s and per-kernel ms.
A=$1; B=$2; shift 2
for t in "$A" "$B"; do
  UBDPT_LIB=$PWD/bidirectional-pathtracing_amd/lib/libubdpt$t.so timeout -k 10 200 python3 bench.py --no-cpu-baseline "$@" > gpurun_out/ab$t.log 2> gpurun_out/ab$t.err || exit 1
  python3 - "$t" <<'PY'
import json,sys
t=sys.argv[1]
d=json.loads(open(f"gpurun_out/ab{t}.log").read().strip().splitlines()[-1])
print("lib%s: %.1f Mpaths/s"%(t, d["value"]), {k: round(v["ms_per_launch"],3) for k,v in d["kernels"].items()})
PY
done

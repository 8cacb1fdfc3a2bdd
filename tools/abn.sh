#!/bin/bash
# A/B of several builds of the library on the GPU box: bash tools/abn.sh "<tag> <tag> ..." [bench args]
# (tag "-" = lib/libubdpt.so, tag "_x" = lib/libubdpt_x.so); prints Mpaths/s and per-kernel ms per launch.
TAGS=$1; shift
for t in $TAGS; do
  [ "$t" = "-" ] && t=""
  UBDPT_LIB=$PWD/bidirectional-pathtracing_amd/lib/libubdpt$t.so timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-other-configs "$@" > gpurun_out/ab$t.log 2> gpurun_out/ab$t.err || exit 1
  python3 - "$t" <<'PY'
import json,sys
t=sys.argv[1]
d=json.loads(open(f"gpurun_out/ab{t}.log").read().strip().splitlines()[-1])
print("lib%-8s %.1f Mpaths/s"%(t, d["value"]), {k.replace("k_",""): round(v["ms_per_launch"],2) for k,v in d["kernels"].items()}, flush=True)
PY
done

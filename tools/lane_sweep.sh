#!/bin/bash
# lanes x passes per batch on the GPU box: bash tools/lane_sweep.sh <config> "<streams>:<passes> ..."
cfg=$1; shift
for sp in $1; do
  s=${sp%%:*}; p=${sp##*:}
  timeout -k 10 200 python3 bench.py --config $cfg --no-cpu-baseline --no-other-configs --steps 8 --streams $s --passes-in-flight $p $LANE_SWEEP_ARGS > gpurun_out/lane_$s_$p.log 2> gpurun_out/lane.err || { tail -3 gpurun_out/lane.err; continue; }
  python3 - "$s" "$p" gpurun_out/lane_$s_$p.log <<'PY'
import json,sys
d=json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print("streams %s passes %s: %.1f Mpaths/s  batch %s" % (sys.argv[1], sys.argv[2], d["value"], d["batch"]), flush=True)
PY
done

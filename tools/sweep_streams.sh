#!/bin/bash
# Sweep of scheduling knobs on the GPU box (C2): bash tools/sweep_streams.sh
set -e
for cfg in "default:" "b256k_s2:--batch-pixel-samples 262143 --streams 2" "b128k_s4:--batch-pixel-samples 131072 --streams 4" "b128k_s8:--batch-pixel-samples 131072 --streams 8" "b64k_s4:--batch-pixel-samples 65536 --streams 4" "b64k_s8:--batch-pixel-samples 65536 --streams 8" "b64k_s16:--batch-pixel-samples 65536 --streams 16" "b32k_s8:--batch-pixel-samples 32768 --streams 8" "b32k_s16:--batch-pixel-samples 32768 --streams 16" "b16k_s16:--batch-pixel-samples 16384 --streams 16"; do
  n=${cfg%%:*}; a=${cfg#*:}
  timeout -k 10 200 python3 bench.py --no-cpu-baseline $a > gpurun_out/sw_$n.log 2> gpurun_out/sw_$n.err
  python3 - "$n" <<'PY'
import json,sys
n=sys.argv[1]
d=json.loads(open(f"gpurun_out/sw_{n}.log").read().strip().splitlines()[-1])
print("%-10s %.1f Mpaths/s"%(n, d["value"]), {k: round(v["ms_per_launch"],3) for k,v in d["kernels"].items()})
PY
done

#!/bin/bash
# Sweep of the shadow-ray step budget on a large scene: bash tools/sweep_budget.sh [config] [passes-per-step] [budgets...]
set -e
CFG=${1:-C3}; PPS=${2:-2}; shift 2 || true
BUDGETS=${@:-0xffffffff 6 8 12 16 24}
for b in $BUDGETS; do
  timeout -k 10 400 python3 bench.py --no-cpu-baseline --config $CFG --passes-per-step $PPS --steps 3 --warmup 1 --shadow-ray-budget $b > gpurun_out/sb_$b.log 2> gpurun_out/sb_$b.err
  python3 - "$b" <<'PY'
import json,sys
n=sys.argv[1]
d=json.loads(open(f"gpurun_out/sb_{n}.log").read().strip().splitlines()[-1])
k=d["kernels"]["k_pair_visible"]
print("budget %-10s %.1f Mpaths/s  visible %.3f ms x %d launches" % (n, d["value"], k["ms_per_launch"], k["launches"]), {a: round(v["ms_per_launch"],3) for a,v in d["kernels"].items() if a != "k_pair_visible"})
PY
done

#!/bin/bash
# GPU parity tests, then the default bench with a one-line summary: bash tools/check_and_bench.sh <tag> [bench args]
tag=${1:-run}; shift
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t_$tag.log 2>&1
rc=$?
tail -3 gpurun_out/t_$tag.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python3 bench.py "$@" > gpurun_out/b_$tag.json 2> gpurun_out/b_$tag.err || exit $?
python3 - "$tag" <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/b_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("%.1f Mpaths/s  %.0f Mrays/s  %.2f rays/sample" % (d["value"], d["mrays_per_s"], d["rays_per_pixel_sample"]))
print({k: round(v["ms_per_launch"],3) for k,v in d["kernels"].items()})
PY

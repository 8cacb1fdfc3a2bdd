#!/bin/bash
# One profiling round on the GPU box, from the repo root:
#   bash tools/profile_round.sh <tag> [bench args...]
# 1. the bench line (with the CPU baseline unless the args say otherwise) -> gpurun_out/<tag>/bench.json
# 2. rocprofv3 --kernel-trace --stats of the same command                  -> gpurun_out/<tag>/stats/
# 3. hardware-counter passes (tools/pmc_passes.sh)                         -> gpurun_out/<tag>/pmc/
set -u
export TMPDIR=/tmp
R=$PWD
T=$R/gpurun_out/$1; shift
TAG=$(basename "$T")
mkdir -p "$T"
timeout -k 10 500 python3 bench.py "$@" > "$T/bench.json" 2> "$T/bench.err" || exit 1
( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$T/stats" -- python3 "$R/bench.py" --no-cpu-baseline --no-other-configs "$@" > "$T/bench_under_rocprof.json" 2> "$T/rocprof.err" ) || exit 2
bash tools/pmc_passes.sh "gpurun_out/$TAG/pmc" "$@" || exit 3
python3 tools/pmc_summary.py "$T/pmc" "$T/pmc/summary.json" || exit 4
echo "profile round $TAG done"

#!/bin/bash
# Counter traffic of one configuration on the GPU box, from the repo root: bash tools/traffic_round.sh <config> <traffic.json to update>
# bench line (for the batch size) + FETCH_SIZE / WRITE_SIZE passes -> entry <config> of the traffic file
set -u
cfg=$1; dst=$2
T=gpurun_out/traffic_$cfg
mkdir -p $T
timeout -k 10 200 python3 bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $T/bench.json 2> $T/bench.err || exit 1
PMC_TRAFFIC_ONLY=1 bash tools/pmc_passes.sh $T/pmc --config $cfg || exit 2
python3 tools/pmc_summary.py $T/pmc $T/pmc/summary.json || exit 3
python3 tools/traffic_from_pmc.py $T/pmc/summary.json $cfg $dst $T/bench.json || exit 4

#!/bin/bash
# Collects rocprofv3 hardware-counter passes of a short bench run, one pass per rocprofv3 run.
# Usage (on the GPU box, from the repo root): bash tools/pmc_passes.sh <outdir> [bench args...]
set -u
export TMPDIR=/tmp
R=$PWD
OUT=$R/$1; shift
mkdir -p "$OUT"
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-other-configs $*"
run_pass() {
  name=$1; shift
  ( cd /tmp && timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$R/bench.py" $ARGS > "$OUT/$name.log" 2>&1 ) || return 1
}
if [ "${PMC_TRAFFIC_ONLY:-0}" = 1 ]; then   # only the two passes the traffic figures need
  run_pass fetch FETCH_SIZE && run_pass write WRITE_SIZE
  echo "pmc passes rc=$?"
  exit 0
fi
run_pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU &&
run_pass sq2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA &&
run_pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum GRBM_GUI_ACTIVE &&
run_pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_CACHE_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum &&
run_pass fetch FETCH_SIZE &&
run_pass write WRITE_SIZE
echo "pmc passes rc=$?"

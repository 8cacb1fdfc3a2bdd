#!/bin/bash
# A/B of tree builders / search radii on the GPU box: bash tools/tree_ab.sh "<config> ..." ; prints Mpaths/s and the traversal work per ray
for cfg in $1; do
for v in karras 8 16 32 64; do
  if [ "$v" = karras ]; then export UBDPT_BVH_BUILDER=karras; unset UBDPT_PLOC_RADIUS; else unset UBDPT_BVH_BUILDER; export UBDPT_PLOC_RADIUS=$v; fi
  timeout -k 10 200 python3 bench.py --config $cfg --no-cpu-baseline --no-other-configs --steps 6 > gpurun_out/tree_$v.log 2> gpurun_out/tree_$v.err || exit 1
  python3 - "$v" "$cfg" <<'PY'
import json,sys
v,cfg=sys.argv[1:3]
d=json.loads(open(f"gpurun_out/tree_{v}.log").read().strip().splitlines()[-1])
w=d["work_per_pixel_sample"]
print("%s %-7s %.1f Mpaths/s  nodes/ray %.2f prims/ray %.2f max stack %d  trace %.2f visible %.2f ms"%(cfg, v, d["value"], w["nodes_per_ray"], w["prims_per_ray"], w["max_stack_entries"], d["kernels"]["k_wf_trace"]["ms_per_launch"], d["kernels"]["k_rays_visible"]["ms_per_launch"]), flush=True)
PY
done
done

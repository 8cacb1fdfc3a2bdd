#!/bin/bash
# Measurement of the persistent ray-tracing waves (library built with -DUBDPT_WAVE_TIMES as lib/libubdpt_wt*.so):
#   bash tools/wave_times.sh "<tag> ..." [bench args]
TAGS=$1; shift
for t in $TAGS; do
UBDPT_LIB=$PWD/bidirectional-pathtracing_amd/lib/libubdpt$t.so timeout -k 10 200 python3 - "$@" <<'PY' || exit 1
import ctypes, json, sys, io, contextlib, os
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-other-configs"] + sys.argv[1:]
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
from importlib import import_module
capi = import_module("bidirectional-pathtracing_amd.capi")
lib = ctypes.CDLL(capi.LIB_PATH)
out = (ctypes.c_ulonglong * (16 * 12 + 12))()
lib.ubdpt_debug_wave_times(out)
print(os.path.basename(capi.LIB_PATH), "%.1f Mpaths/s" % d["value"], "trace ms/launch %.3f" % d["kernels"]["k_wf_trace"]["ms_per_launch"],
      "nodes/ray %.2f" % d["work_per_pixel_sample"]["nodes_per_ray"])
print("  depth  launches  Mrays  waves/launch(work)  life_us  longest_us(max over launches)  after_dry_us  claim_wait_us  iters/wave  node_iters  leaf_iters  us/iter  rays/wave")
for dep in range(16):
    g = [out[dep * 12 + i] for i in range(12)]
    w, life, tail, iters, steps, empty, longest, claim, launches, rays, leaf = g[:11]
    if not launches or not w: continue
    print("  %5d  %8d  %5.2f  %18.0f  %7.1f  %10.1f  %12.1f  %13.1f  %10.0f  %10.0f  %10.0f  %7.2f  %9.0f"
          % (dep, launches, rays / launches / 1e6, w / launches, life / w / 100.0, longest / 100.0, tail / w / 100.0, claim / w / 100.0,
             iters / w, steps / w, leaf / w, life / iters / 100.0, rays / w))
ph = [out[16 * 12 + k] for k in range(8)]
tot = float(sum(ph)) or 1.0
print("  loop phases: arrivals %.1f %%, feeding %.1f %%, node fetch wait %.1f %%, node arithmetic %.1f %%, primitive fetch wait %.1f %%, local ray %.1f %%, primitive test %.1f %%, gate / pop / store %.1f %%"
      % tuple(100.0 * x / tot for x in ph))
ln = [out[16 * 12 + 8 + k] for k in range(4)]
tl = float(sum(ln)) or 1.0
print("  lanes at the node step: idle %.1f %%, waiting for their ray %.1f %%, at an inner node %.1f %%, at a leaf %.1f %%" % tuple(100.0 * x / tl for x in ln))
PY
done

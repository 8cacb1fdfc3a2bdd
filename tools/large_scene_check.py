#!/usr/bin/env python3
"""Large-scene check on the GPU: LBVH against the linear scan (both on the device, fp64 leaf tests in
reference order) on a seeded ray batch, build time, traversal statistics and throughput.

  python tools/large_scene_check.py C3 [n_rays] [passes]
"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("bidirectional-pathtracing_amd")
capi, scenes = pkg.capi, pkg.scenes
from util import ray_batch, segment_batch  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 14
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 4
t = time.perf_counter()
scene, params, spp = scenes.make_config(name)
out = {"config": name, "faces": int(scene.num_faces), "scene_gen_s": time.perf_counter() - t}
print(out, flush=True)
rays = ray_batch(n_rays, seed=42)
segs = segment_batch(n_rays, seed=43)
res = {}
for accel in (0, 1):
    opt = capi.Options()
    opt.accel = accel
    opt.collect_stats = 1
    opt.passes_in_flight = 1
    opt.streams = 1
    t = time.perf_counter()
    with capi.Context(scene, params, opt) as ctx:
        out[f"create_s_accel{accel}"] = time.perf_counter() - t
        t = time.perf_counter()
        h = ctx.intersect_batch(rays)
        v = ctx.visibility_batch(segs)
        out[f"probe_s_accel{accel}"] = time.perf_counter() - t
        res[accel] = (h, v)
        if accel == 0:
            st = ctx.stats()
            out["build_seconds"] = st["build_seconds"]
            for k in st["kernels"]:
                if k["name"] == "k_intersect_batch":
                    out["nodes_per_closest_ray"] = k["nodes_visited"] / n_rays
                    out["prims_per_closest_ray"] = k["prims_tested"] / n_rays
    print(out, flush=True)
h0, v0 = res[0]
h1, v1 = res[1]
same = (np.array_equal(h0["hit"], h1["hit"]) and np.array_equal(h0["object"], h1["object"]) and np.array_equal(h0["face"], h1["face"]))
m = h1["hit"] == 1
bits = all(np.array_equal(h0[f][m].view(np.uint64), h1[f][m].view(np.uint64)) for f in ("d", "pos", "Ng", "Ns", "Ts", "Bs", "tex_u", "tex_v"))
out["hits"] = int(m.sum())
out["lbvh_equals_linear_identity"] = bool(same)
out["lbvh_equals_linear_bits"] = bool(bits)
out["visibility_equal"] = bool(np.array_equal(v0, v1))
print(out, flush=True)
# throughput, default options
opt = capi.Options()
with capi.Context(scene, params, opt) as ctx:
    ctx.render_passes(1 << 30, 1, 0x5EED)
    ctx.upload_sum(None)
    ctx.reset_stats()
    t = time.perf_counter()
    ctx.render_passes(0, passes, 0x5EED)
    dt = time.perf_counter() - t
    img = ctx.read_sum()
    st = ctx.stats()
npix = int(params.img_res_x * params.img_res_y)
out["mpaths_per_s"] = npix * passes / dt / 1e6
out["finite"] = bool(np.isfinite(img).all())
out["mean_radiance"] = [float(x) for x in img.reshape(-1, 3).mean(0) / passes]
out["kernel_ms_per_launch"] = {k["name"]: k["ms"] / max(1, k["launches"]) for k in st["kernels"]}
print(json.dumps(out), flush=True)

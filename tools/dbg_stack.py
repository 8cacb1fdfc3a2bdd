import sys, numpy as np
sys.path.insert(0, 'tests')
from util import capi, scenes
import importlib
t = importlib.import_module("test_gpu_parity")
scene = t.stacked_triangles_scene()
rays = np.zeros(64, capi.RAY_DTYPE)
rays["origin"] = (0.0, 0.0, -1.0)
rays["dir"] = (0.0, 0.0, 1.0)
for cap in (0, 6):
    opt = capi.Options(); opt.traversal_stack_entries = cap
    with capi.Context(scene, scenes.make_params(8, 8), opt) as ctx:
        try:
            h = ctx.intersect_batch(rays)
            print(cap, "ok", h["face"][:4], h["d"][:4])
        except Exception as e:
            print(cap, "raised", e)
        s = ctx.stats()
        print([ (k["name"], k["nodes_visited"], k["prims_tested"], k["max_stack_entries"]) for k in s["kernels"]])

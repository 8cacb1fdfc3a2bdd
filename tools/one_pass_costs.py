"""Costs of the pieces of one reference-shaped renderPass on C2 512x512: one pass rendered alone, clearing the
device sum, reading it back (with and without adding into the caller's buffer), a host copy of the buffer."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("bidirectional-pathtracing_amd")
capi, scenes = pkg.capi, pkg.scenes
import numpy as np
sc, params, _ = scenes.make_config("C2")
with capi.Context(sc, params) as ctx:
    for k in range(4): ctx.render_passes(k, 1, 1)
    n = 64
    t = time.perf_counter()
    for k in range(n): ctx.render_passes(100 + k, 1, 1)
    dt = (time.perf_counter() - t) / n
    print("render 1 pass sync: %.3f ms" % (dt * 1e3))
    t = time.perf_counter()
    for k in range(n): ctx.upload_sum(None)
    print("clear: %.3f ms" % ((time.perf_counter() - t) / n * 1e3))
    buf = np.zeros((512, 512, 3))
    t = time.perf_counter()
    for k in range(n): ctx.read_sum(buf, add_into=True)
    print("read_sum add_into: %.3f ms" % ((time.perf_counter() - t) / n * 1e3))
    t = time.perf_counter()
    for k in range(n): ctx.read_sum(buf)
    print("read_sum: %.3f ms" % ((time.perf_counter() - t) / n * 1e3))
    t = time.perf_counter()
    for k in range(n): b2 = buf.copy()
    print("host copy 6.3MB: %.3f ms" % ((time.perf_counter() - t) / n * 1e3))

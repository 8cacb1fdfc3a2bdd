#!/usr/bin/env python3
"""PCIe-inclusive rate of the reference-shaped interface: HipEngine::renderPass hands a HOST sum
buffer to the renderer every pass (device clear, one pass, read-back, add on the host), against
HipEngine::renderPasses, which keeps the sum on the device. C2 at 512x512."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("bidirectional-pathtracing_amd")
hostapi = importlib.import_module("bidirectional-pathtracing_amd.hostapi")

sc, params, _ = pkg.scenes.make_config("C2")
npix = int(params.img_res_x * params.img_res_y)
with hostapi.Engine(sc) as eng:
    assert eng.new_rendering(params) == 0
    for _ in range(4):
        eng.render_pass()
    t = time.perf_counter()
    n = 32
    for _ in range(n):
        assert eng.render_pass() == 0
    dt = time.perf_counter() - t
    print(f"pass-by-pass through the host buffer: {npix * n / dt / 1e6:.1f} Mpaths/s ({dt / n * 1e3:.2f} ms per pass)")
    eng.render_passes(16)
    t = time.perf_counter()
    assert eng.render_passes(256) == 0
    dt = time.perf_counter() - t
    print(f"batched, sum resident on the device: {npix * 256 / dt / 1e6:.1f} Mpaths/s")

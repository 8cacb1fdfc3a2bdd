#!/usr/bin/env python3
"""PCIe-inclusive rate of the reference-shaped interface: HipEngine::renderPass hands a HOST sum
buffer to the renderer every pass (device clear, one pass, read-back, add on the host), against
HipEngine::renderPasses, which keeps the sum on the device: python tools/host_pass_rate.py [config, default C2] [pass_slots]."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("bidirectional-pathtracing_amd")
hostapi = importlib.import_module("bidirectional-pathtracing_amd.hostapi")

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
sc, params, _ = pkg.scenes.make_config(cfg)
npix = int(params.img_res_x * params.img_res_y)
opt = None
if len(sys.argv) > 2:  # per-pass buffers of the engine's renderer (two windows of half as many passes)
    opt = pkg.capi.Options()
    opt.pass_slots = int(sys.argv[2])
with hostapi.Engine(sc, options=opt) as eng:
    assert eng.new_rendering(params) == 0
    # passes are served from windows rendered ahead: warm up over two windows, then time four, so that the timed
    # region neither consumes work done before it nor leaves work behind
    win = max(1, min(32, (1 << 24) // npix)) if opt is None else max(1, opt.pass_slots // 2)
    for _ in range(2 * win):
        eng.render_pass()
    t = time.perf_counter()
    n = 4 * win
    for _ in range(n):
        assert eng.render_pass() == 0
    dt = time.perf_counter() - t
    print(f"pass-by-pass through the host buffer (HipEngine, windows of {win}): {npix * n / dt / 1e6:.1f} Mpaths/s ({dt / n * 1e3:.2f} ms per pass)")
    nb = 256 if npix < (1 << 20) else 32
    eng.render_passes(16 if npix < (1 << 20) else 4)
    t = time.perf_counter()
    assert eng.render_passes(nb) == 0
    dt = time.perf_counter() - t
    print(f"batched, sum resident on the device: {npix * nb / dt / 1e6:.1f} Mpaths/s")

# the loop of the reference's engine around the URenderer subclass of host/reference_adapter/ (same calls, timed in C++)
lib = hostapi.load_library()
import ctypes as C
for copy, what in ((1, "with the engine's own copy of the previous sum (uengine.cpp:153-156)"), (0, "renderer alone")):
    n = 64 if npix < (1 << 20) else 32
    sec = lib.uhip_adapter_pass_rate(C.byref(sc.desc), C.byref(params), 0, n // 2, n, 8, copy)
    print(f"adapter loop, {what}: {npix * n / sec / 1e6:.1f} Mpaths/s ({sec / n * 1e3:.2f} ms per pass)")

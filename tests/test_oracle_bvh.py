"""Mode B of the CPU oracle (a bounding-volume hierarchy per mesh, used as the CPU baseline of the mesh
workloads) against its Mode A (the reference's scans over every face, uapp/mesh.cpp:123-137, :170-180):
the tree may only change which faces are tested, never which hit is reported."""
import numpy as np
import pytest

import oracle_py
from util import capi, ray_batch, scenes, segment_batch

SCENES = {
    "C1": (scenes.cornell_c1, 1 << 14),
    "tiny": (scenes.tiny_scene, 1 << 14),
    "mesh6k": (lambda: scenes.mesh_scene(6000, seed=3, ratio=1.0), 1 << 13),
    "mesh30k_mixed": (lambda: scenes.mesh_scene(30000, seed=4, ratio=1.0, mixed=True), 1 << 11),
}


@pytest.mark.parametrize("name", list(SCENES))
def test_tree_reports_the_hits_of_the_scan(name):
    factory, n = SCENES[name]
    scene = factory()
    params = scenes.make_params(32, 32)
    a = oracle_py.Oracle(scene, params)
    b = oracle_py.Oracle(scene, params).set_accel(1)
    rays, segs = ray_batch(n, seed=42), segment_batch(n, seed=43)
    ha, hb = a.intersect_batch(rays), b.intersect_batch(rays)
    for f in capi.HIT_DTYPE.names:
        assert np.array_equal(np.ascontiguousarray(ha[f]).view(np.uint8), np.ascontiguousarray(hb[f]).view(np.uint8)), f
    assert ha["hit"].sum() > n // 2
    va, vb = a.visibility_batch(segs), b.visibility_batch(segs)
    assert np.array_equal(va, vb) and 0 < va.sum() < n


def test_tree_renders_the_image_of_the_scan_with_fewer_tests():
    scene = scenes.mesh_scene(6000, seed=3, ratio=1.0)
    params = scenes.make_params(24, 24)
    a = oracle_py.Oracle(scene, params)
    b = oracle_py.Oracle(scene, params).set_accel(1)
    ia, ca = a.render_passes(0, 2, 0x5EED, counters=True)
    ib, cb = b.render_passes(0, 2, 0x5EED, counters=True)
    assert np.array_equal(ia, ib) and ia.sum() > 0
    for k in ("pixel_samples", "closest_hit_rays", "shadow_rays", "eye_vertices", "light_vertices", "connections", "splats"):
        assert ca[k] == cb[k], k
    assert cb["nodes_visited"] > 0 and ca["nodes_visited"] == 0
    assert cb["prims_tested"] * 50 < ca["prims_tested"]
    # switching back gives the scan again
    b.set_accel(0)
    ic, cc = b.render_passes(0, 2, 0x5EED, counters=True)
    assert np.array_equal(ia, ic) and cc["prims_tested"] == ca["prims_tested"]

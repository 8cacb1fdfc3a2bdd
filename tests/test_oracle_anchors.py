"""Statistical anchors and invariants of the CPU oracle. Runs without a GPU.

The only outputs of the real reference renderer available to this repository are the aggregate
figures of a probe run of its unmodified sources on the procedural Cornell box (scene "C1":
256 x 256, 16 passes): mean radiance, rays per pixel-sample and mean subpath lengths. The oracle
must reproduce them within Monte-Carlo noise; everything else here is a property of the algorithm
(pass additivity, independence of the thread count, partition of the passes over ranks)."""
import numpy as np
import pytest

from util import capi, scenes, rel_mse

import oracle_py

SEED = 0x5EED

# probe of the reference on the C1 box: 256^2, 16 passes
REF_MEAN_RADIANCE = (0.752, 0.722, 0.684)
REF_RAYS_PER_PIXEL_SAMPLE = 13.4
REF_MEAN_EYE_LEN = 3.02
REF_MEAN_LIGHT_LEN = 2.61


@pytest.fixture(scope="module")
def c1_render():
    sc, params, _ = scenes.make_config("C1")
    orc = oracle_py.Oracle(sc, params)
    img, cn = orc.render_passes(0, 16, SEED, num_threads=8, counters=True)
    return img / 16.0, cn


def test_mean_radiance_matches_reference_probe(c1_render):
    img, _ = c1_render
    mean = img.reshape(-1, 3).mean(0)
    assert np.all(np.isfinite(img))
    for got, ref in zip(mean, REF_MEAN_RADIANCE):
        assert abs(got - ref) / ref < 0.01, (mean, REF_MEAN_RADIANCE)


def test_ray_and_vertex_statistics_match_reference_probe(c1_render):
    _, cn = c1_render
    ps = cn["pixel_samples"]
    assert ps == 256 * 256 * 16
    rays = (cn["closest_hit_rays"] + cn["shadow_rays"]) / ps
    assert abs(rays - REF_RAYS_PER_PIXEL_SAMPLE) / REF_RAYS_PER_PIXEL_SAMPLE < 0.01, rays
    assert abs(cn["eye_vertices"] / ps - REF_MEAN_EYE_LEN) < 0.02
    assert abs(cn["light_vertices"] / ps - REF_MEAN_LIGHT_LEN) < 0.02
    assert cn["capped_subpaths"] == 0


def small(name="C2", w=48, h=48):
    sc, params, _ = scenes.make_config(name)
    params.img_res_x, params.img_res_y = w, h
    return sc, params


def test_passes_are_additive_and_thread_count_does_not_matter():
    sc, params = small()
    orc = oracle_py.Oracle(sc, params)
    whole = orc.render_passes(0, 6, SEED, num_threads=1)
    parts = orc.render_passes(0, 2, SEED, num_threads=3) + orc.render_passes(2, 4, SEED, num_threads=8)
    # same samples; only the order of the additions into a pixel differs
    assert np.allclose(whole, parts, rtol=1e-12, atol=1e-13)
    again = orc.render_passes(0, 6, SEED, num_threads=1)
    assert np.array_equal(whole, again)


def test_round_robin_partition_reproduces_the_full_render():
    sc, params = small("C1")
    orc = oracle_py.Oracle(sc, params)
    whole = orc.render_passes(0, 8, SEED, num_threads=4)
    for G in (2, 4):
        total = np.zeros_like(whole)
        for g in range(G):
            total += orc.render_passes(g, 8 // G, SEED, pass_stride=G, num_threads=4)
        assert np.allclose(whole, total, rtol=1e-12, atol=1e-13)


def test_seeds_give_independent_estimates_of_the_same_image():
    sc, params = small("C1", 32, 32)
    orc = oracle_py.Oracle(sc, params)
    a = orc.render_passes(0, 64, 1, num_threads=8) / 64
    b = orc.render_passes(0, 64, 2, num_threads=8) / 64
    assert not np.array_equal(a, b)
    assert rel_mse(a, b) < 0.05
    assert np.allclose(a.reshape(-1, 3).mean(0), b.reshape(-1, 3).mean(0), rtol=0.03)


def test_strata_follow_the_pass_index():
    # pixel_subdiv = 2: pass p samples stratum p % 4 (uengine/ubdptrenderer.cpp:95-96); a full cycle of
    # strata has lower variance than four draws of one stratum would, and each pass differs
    sc, params = small("C1", 24, 24)
    params.pixel_subdiv = 2
    orc = oracle_py.Oracle(sc, params)
    imgs = [orc.render_passes(p, 1, SEED) for p in range(4)]
    assert all(not np.array_equal(imgs[0], imgs[k]) for k in (1, 2, 3))
    assert np.isfinite(sum(imgs)).all()


def test_path_cap_only_truncates():
    sc, params = small("C2", 32, 32)
    full = oracle_py.Oracle(sc, params, 32)
    capped = oracle_py.Oracle(sc, params, 4)
    a, ca = full.render_passes(0, 4, SEED, num_threads=4, counters=True)
    b, cb = capped.render_passes(0, 4, SEED, num_threads=4, counters=True)
    assert cb["capped_subpaths"] > 0 and ca["capped_subpaths"] == 0
    assert cb["eye_vertices"] <= ca["eye_vertices"] and cb["light_vertices"] <= ca["light_vertices"]
    assert np.isfinite(b).all()


def test_trace_pixel_agrees_with_the_image():
    sc, params = small("C2", 16, 16)
    orc = oracle_py.Oracle(sc, params)
    img = orc.render_passes(3, 1, SEED)
    acc = np.zeros_like(img)
    for py in range(16):
        for px in range(16):
            eye, light, I, splats = orc.trace_pixel(3, SEED, px, py)
            assert len(eye) >= 1 and eye[0]["object"] == -1      # lens vertex
            assert np.isinf(eye[0]["p_eye_A"])                   # lens area is zero in the reference
            acc[py, px] += I
            for s in splats:
                acc[int(s[1]), int(s[0])] += s[2:5]
    assert np.allclose(acc, img, rtol=1e-12, atol=1e-14)

"""The committed fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py from the CPU
oracle) against the oracle (no GPU needed) and against the HIP backend (marked gpu).

The fixtures are oracle output, not reference output: the reference renderer cannot be built here.
They pin the oracle against silent drift and let the backend be checked where only this repository
exists."""
import numpy as np
import pytest

from util import GOLDEN_SCENES, bsdf_queries, capi, load_golden, ray_batch, scenes, segment_batch

NAMES = list(GOLDEN_SCENES)


def build(name):
    g = load_golden(f"scene_{name}.npz")
    sc = GOLDEN_SCENES[name]()
    sc.desc.camera.ratio = 1.0
    params = scenes.make_params(int(g["image_w"]), int(g["image_h"]))
    rays = ray_batch(int(g["n_rays"]), seed=int(g["ray_seed"]))
    segs = segment_batch(int(g["n_segs"]), seed=int(g["seg_seed"]))
    return g, sc, params, rays, segs


def check_hits(g, hits):
    assert np.array_equal(hits["hit"], g["hit"])
    assert np.array_equal(hits["object"], g["object"])
    assert np.array_equal(hits["face"], g["face"])
    m = g["hit"] == 1
    for f in ("d", "Ns", "tex_u", "tex_v"):   # bit patterns: NaN frames of sphere equators compare equal too
        assert np.array_equal(hits[f][m].view(np.uint64), g[f][m].view(np.uint64)), f


# ---- oracle vs fixtures (CPU) -------------------------------------------------------------------------
@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_scene_fixture(name):
    import oracle_py
    g, sc, params, rays, segs = build(name)
    orc = oracle_py.Oracle(sc, params)
    check_hits(g, orc.intersect_batch(rays))
    assert np.array_equal(np.packbits(orc.visibility_batch(segs)), g["visible"])
    img, cn = orc.render_passes(0, int(g["image_passes"]), int(g["image_seed"]), num_threads=4, counters=True)
    # multi-threaded accumulation order is not fixed: compare to rounding noise, counters exactly
    assert np.allclose(img, g["image_sum"], rtol=1e-12, atol=1e-13)
    got = [cn[k] for k in ("closest_hit_rays", "shadow_rays", "eye_vertices", "light_vertices", "connections", "splats")]
    assert got == list(g["counters"])


def test_oracle_reproduces_bsdf_fixture():
    import oracle_py
    g = load_golden("bsdf.npz")
    q = bsdf_queries(int(g["n"]), seed=int(g["seed"]))
    assert np.array_equal(oracle_py.bsdf_eval(q)["f"], g["eval_f"])
    assert np.array_equal(oracle_py.bsdf_pdf(q)["pdf"], g["pdf"])
    s = oracle_py.bsdf_scatter(q)
    ok = g["scatter_ok"] == 1
    assert np.array_equal(s["ok"] == 1, ok)
    assert np.array_equal(s["dir"][ok], g["scatter_dir"][ok]) and np.array_equal(s["pdf"][ok], g["scatter_pdf"][ok])
    assert np.array_equal(s["f"][ok], g["scatter_f"][ok]) and np.array_equal(s["specular"][ok], g["scatter_specular"][ok])


def test_oracle_reproduces_path_dumps():
    import oracle_py
    g = load_golden("paths_C2.npz")
    sc, params, _ = scenes.make_config("C2")
    params.img_res_x = params.img_res_y = 32
    orc = oracle_py.Oracle(sc, params)
    k = 0
    while f"p{k}_where" in g:
        px, py, pas = (int(x) for x in g[f"p{k}_where"])
        eye, light, I, splats = orc.trace_pixel(pas, 0x5EED, px, py)
        assert eye.tobytes() == g[f"p{k}_eye"].tobytes() and light.tobytes() == g[f"p{k}_light"].tobytes()
        assert np.array_equal(I, g[f"p{k}_I"]) and np.array_equal(splats, g[f"p{k}_splats"])
        k += 1
    assert k == 6


# ---- HIP backend vs fixtures (GPU) -----------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("accel", [0, 1])
@pytest.mark.parametrize("name", NAMES)
def test_backend_reproduces_scene_fixture(name, accel):
    g, sc, params, rays, segs = build(name)
    opt = capi.Options()
    opt.accel = accel
    opt.collect_stats = 1
    with capi.Context(sc, params, opt) as ctx:
        check_hits(g, ctx.intersect_batch(rays))
        assert np.array_equal(np.packbits(ctx.visibility_batch(segs)), g["visible"])
        ctx.reset_stats()
        ctx.render_passes(0, int(g["image_passes"]), int(g["image_seed"]))
        img = ctx.read_sum()
        st = ctx.stats()
    ref = g["image_sum"]
    # same uniforms on both sides; a handful of paths branch differently where the device's sin / cos
    # differ from glibc's in the last bit, so a few pixels may differ and the counters by a few units
    bad = np.abs(img - ref) > 1e-9 * (np.abs(ref) + 1e-3)
    assert bad.any(axis=2).mean() <= 0.002  # observed 0; four pixels of a 48 x 48 image may branch on a last-bit sin / cos difference
    assert abs(img.mean() - ref.mean()) / ref.mean() < 5e-3
    got = np.array([st[k] for k in ("closest_hit_rays", "shadow_rays", "eye_vertices", "light_vertices", "connections", "splats")],
                   np.float64)
    tol = 2e-3 * g["counters"] + 2
    keep = np.array([False, False, True, True, True, True])
    assert np.all(np.abs(got - g["counters"])[keep] <= tol[keep])
    # shadow rays: the reference (and the oracle) trace the visibility ray first and test the geometric-normal
    # cosines afterwards (ubdptrenderer.cpp:609-626); the backend tests the cosines first and skips
    # the ray of a connection that would be rejected anyway, so it traces fewer, never more
    assert got[1] <= g["counters"][1] + tol[1] and got[1] > 0.5 * g["counters"][1]
    # closest-hit rays: the reference intersects first and plays Russian roulette afterwards
    # (ubdptrenderer.cpp:340-356); the backend decides the roulette (a counter-based draw) before it
    # traces, so a subpath that loses is not traced again: fewer rays, the same vertices
    assert got[0] <= g["counters"][0] + tol[0] and got[0] > 0.8 * g["counters"][0]


@pytest.mark.gpu
def test_backend_reproduces_bsdf_fixture():
    g = load_golden("bsdf.npz")
    q = bsdf_queries(int(g["n"]), seed=int(g["seed"]))
    sc, params, _ = scenes.make_config("C1")
    params.img_res_x = params.img_res_y = 8
    with capi.Context(sc, params) as ctx:
        assert np.array_equal(ctx.bsdf_eval(q)["f"], g["eval_f"])
        assert np.array_equal(ctx.bsdf_pdf(q)["pdf"], g["pdf"])
        s = ctx.bsdf_scatter(q)
    ok = g["scatter_ok"] == 1
    assert np.array_equal(s["ok"] == 1, ok)
    # sampled directions go through sin / cos: device and glibc may differ in the last bits
    assert np.allclose(s["dir"][ok], g["scatter_dir"][ok], rtol=0, atol=4e-16)
    assert np.allclose(s["pdf"][ok], g["scatter_pdf"][ok], rtol=1e-14)
    assert np.allclose(s["f"][ok], g["scatter_f"][ok], rtol=1e-14)
    assert np.array_equal(s["specular"][ok], g["scatter_specular"][ok])

"""Parity of the HIP backend (through the C ABI) against the CPU oracle on the same seeded inputs.

Bars: hit / object / face index, visibility bits and every derived scene quantity bit-exact;
d, positions and normals bit-exact as well (fp64, same operation order, no FMA contraction);
images: the same counter-based random numbers drive both sides, so pixels agree to rounding noise
except for the few paths whose device sin/cos differ from glibc in the last bit and therefore branch
differently; the tolerance is stated in each test.
"""
import numpy as np
import pytest

from util import bsdf_queries, capi, ray_batch, rel_mse, scenes, segment_batch

pytestmark = pytest.mark.gpu

SEED = 0x5EED


def small_mesh_scene():
    return scenes.mesh_scene(6000, seed=3, ratio=1.0)


SCENES = {
    "C1": scenes.cornell_c1,
    "C2": scenes.cornell_c2,
    "tiny": scenes.tiny_scene,
    "mesh6k": small_mesh_scene,
}


@pytest.fixture(scope="module")
def oracle_mod():
    import oracle_py
    return oracle_py


def make_pair(oracle_mod, name, w=64, h=64, accel=0, stats=0, maxv=32, pif=0, pipeline=0):
    scene = SCENES[name]()
    scene.desc.camera.ratio = w / h
    params = scenes.make_params(w, h)
    opt = capi.Options()
    opt.accel = accel
    opt.collect_stats = stats
    opt.max_path_vertices = maxv
    opt.passes_in_flight = pif
    opt.pipeline = pipeline
    ctx = capi.Context(scene, params, opt)
    orc = oracle_mod.Oracle(scene, params, maxv)
    return scene, ctx, orc


@pytest.mark.parametrize("name", list(SCENES))
def test_flattening_matches_oracle_bitwise(oracle_mod, name):
    scene, ctx, orc = make_pair(oracle_mod, name)
    for i in range(scene.n_objects):
        a, b = ctx.object_info(i), orc.object_info(i)
        assert np.array_equal(a[0], b[0]), f"invW of object {i}"
        assert a[1] == b[1] and a[2] == b[2], f"area/probability of object {i}"
        assert np.array_equal(a[3], b[3]), f"bounding sphere of object {i}"
    ca, cb = ctx.camera_info(), orc.camera_info()
    assert np.array_equal(ca[0], cb[0]) and np.array_equal(ca[1], cb[1]) and ca[2] == cb[2]
    ctx.close()


@pytest.mark.parametrize("accel", [0, 1])
@pytest.mark.parametrize("name", list(SCENES))
def test_intersection_bit_exact(oracle_mod, name, accel):
    scene, ctx, orc = make_pair(oracle_mod, name, accel=accel)
    n = 1 << 16 if name != "mesh6k" else 1 << 14
    rays = ray_batch(n, seed=42)
    got = ctx.intersect_batch(rays)
    ref = orc.intersect_batch(rays)
    assert np.array_equal(got["hit"], ref["hit"])
    assert np.array_equal(got["object"], ref["object"])
    assert np.array_equal(got["face"], ref["face"])
    hit = ref["hit"] == 1
    assert hit.sum() > n // 2
    # bit-exact: compare the raw bytes of every double field of the hits (NaN frames on sphere equators included)
    for f in ("d", "pos", "Ng", "Ns", "Ts", "Bs", "tex_u", "tex_v"):
        assert np.array_equal(got[f][hit].view(np.uint64), ref[f][hit].view(np.uint64)), f
    ctx.close()


@pytest.mark.parametrize("accel", [0, 1])
@pytest.mark.parametrize("name", list(SCENES))
def test_visibility_bit_exact(oracle_mod, name, accel):
    scene, ctx, orc = make_pair(oracle_mod, name, accel=accel)
    n = 1 << 16 if name != "mesh6k" else 1 << 14
    segs = segment_batch(n, seed=43)
    got = ctx.visibility_batch(segs)
    ref = orc.visibility_batch(segs)
    assert np.array_equal(got, ref)
    assert 0 < ref.sum() < n
    ctx.close()


def test_intersection_edge_cases(oracle_mod):
    """Rays through shared edges / vertices of the quads, grazing sphere tangents, origins on surfaces,
    axis-parallel directions (zero direction components) and rays that start outside pointing away."""
    scene, ctx, orc = make_pair(oracle_mod, "C1")
    rays = []
    for x in (-1.0, -0.5, 0.0, 0.25, 1.0):          # towards the floor diagonal / edges / corners
        for z in (-1.0, 0.0, 1.0):
            rays.append(((0.0, 0.5, 0.0), (x, -1.5, z)))
            rays.append(((x, 0.0, z), (0.0, -1.0, 0.0)))   # axis parallel, may start on a wall plane
            rays.append(((x, 0.0, z), (0.0, 0.0, 1.0)))
    c = np.array((-0.4, -0.65, 0.3)); r = 0.35
    for k in range(16):                              # tangent rays to sphere A
        ang = 2 * np.pi * k / 16
        p = c + r * np.array((np.cos(ang), np.sin(ang), 0.0))
        rays.append((tuple(p - np.array((0, 0, 2.0))), (0.0, 0.0, 1.0)))
    rays.append(((0, 0, -5), (0, 0, -1)))            # pointing away from everything
    rays.append(((-0.4, -0.65, 0.3), (0, 1, 0)))     # from the centre of sphere A
    rays.append(((0, -1.0, 0), (1, 0, 0)))           # travelling inside the floor plane
    arr = np.zeros(len(rays), capi.RAY_DTYPE)
    for i, (o, d) in enumerate(rays):
        arr["origin"][i] = o
        arr["dir"][i] = d
    for accel_ctx in (ctx,):
        got = accel_ctx.intersect_batch(arr)
        ref = orc.intersect_batch(arr)
        assert np.array_equal(got["hit"], ref["hit"])
        assert np.array_equal(got["object"], ref["object"])
        assert np.array_equal(got["face"], ref["face"])
        hit = ref["hit"] == 1
        assert np.array_equal(got["d"][hit].view(np.uint64), ref["d"][hit].view(np.uint64))
    ctx.close()


def test_empty_batches_and_bad_arguments(oracle_mod):
    scene, ctx, orc = make_pair(oracle_mod, "C1", w=8, h=8)
    assert ctx.intersect_batch(np.zeros(0, capi.RAY_DTYPE)).shape == (0,)
    assert ctx.visibility_batch(np.zeros(0, capi.SEGMENT_DTYPE)).shape == (0,)
    lib = capi.load_library()
    assert lib.ubdpt_intersect_batch(ctx.handle, None, 5, None) == capi.INVALID_ARG
    assert lib.ubdpt_get_object_info(ctx.handle, 10 ** 6, None, None, None, None) == capi.INVALID_ARG
    st, done = ctx.render_passes(0, 0, SEED)
    assert st == capi.SUCCESS and done == 0
    assert np.all(ctx.read_sum() == 0)
    ctx.close()
    # a scene whose mesh range is out of bounds is rejected
    bad = scenes.cornell_c1()
    bad.objects[0].num_faces = 10 ** 6
    with pytest.raises(capi.UbdptError) as e:
        capi.Context(bad, scenes.make_params(8, 8))
    assert e.value.status == capi.INVALID_SCENE


def test_bsdf_parity(oracle_mod):
    scene, ctx, orc = make_pair(oracle_mod, "C1", w=8, h=8)
    q = bsdf_queries(20000)
    for dev, ref in ((ctx.bsdf_eval(q), oracle_mod.bsdf_eval(q)), (ctx.bsdf_pdf(q), oracle_mod.bsdf_pdf(q))):
        assert np.array_equal(dev["f"].view(np.uint64), ref["f"].view(np.uint64))
        assert np.array_equal(dev["pdf"].view(np.uint64), ref["pdf"].view(np.uint64))
    dev, ref = ctx.bsdf_scatter(q), oracle_mod.bsdf_scatter(q)
    assert np.array_equal(dev["ok"], ref["ok"]) and np.array_equal(dev["specular"], ref["specular"])
    assert np.array_equal(dev["pdf"].view(np.uint64), ref["pdf"].view(np.uint64))
    assert np.array_equal(dev["f"].view(np.uint64), ref["f"].view(np.uint64))
    # sampled directions go through sin/cos: device libm vs glibc, a few ulp
    assert np.allclose(dev["dir"], ref["dir"], rtol=0, atol=1e-14)
    spec = ref["specular"] == 1
    assert np.array_equal(dev["dir"][spec].view(np.uint64), ref["dir"][spec].view(np.uint64))
    ctx.close()


# Largest fraction of pixels that may deviate from the oracle by more than 1e-9 relative in a path-by-path
# comparison. Observed over every such test of this suite: 0.000 % (the device's sin / cos differ from glibc's in
# the last bit for a few arguments, which can send a path another way once in some 10^5 samples). The allowance is
# four pixels of a 48 x 48 image.
BAD_PIXELS_MAX = 0.002
OBSERVED = []


def image_agreement(got, ref, passes):
    """Fraction of pixels whose per-pass mean differs by more than 1e-9 (relative to max(1e-3, value))."""
    g = got / passes; r = ref / passes
    err = np.abs(g - r) / np.maximum(np.abs(r), 1e-3)
    bad = np.any(err > 1e-9, axis=2)
    OBSERVED.append(float(bad.mean()))
    print(f"pixels off by > 1e-9: {bad.mean() * 100:.4f} % (largest so far {max(OBSERVED) * 100:.4f} %)")
    return float(bad.mean()), rel_mse(g, r)


@pytest.mark.parametrize("accel", [0, 1])
@pytest.mark.parametrize("pipeline", [0, 1])
@pytest.mark.parametrize("name,passes", [("C1", 4), ("C2", 4), ("tiny", 3), ("mesh6k", 2)])
def test_image_matches_oracle_path_by_path(oracle_mod, name, passes, accel, pipeline):
    """Same seed => same paths. Tolerance: at most BAD_PIXELS_MAX (0.2 %) of the pixels may deviate by more than
    1e-9 relative (paths that branch differently after a last-bit sin/cos difference, plus the pixels their
    splats land on), relMSE of the whole image <= 2e-3, channel means within 0.5 %."""
    w = h = 48
    scene, ctx, orc = make_pair(oracle_mod, name, w=w, h=h, accel=accel, stats=1, pipeline=pipeline)
    st, done = ctx.render_passes(0, passes, SEED)
    assert st == capi.SUCCESS and done == passes
    got = ctx.read_sum()
    ref, cn = orc.render_passes(0, passes, SEED, num_threads=4, counters=True)
    assert np.isfinite(got).all() == np.isfinite(ref).all()
    fin = np.isfinite(ref).all(axis=2) & np.isfinite(got).all(axis=2)
    bad, rm = image_agreement(np.where(fin[..., None], got, 0), np.where(fin[..., None], ref, 0), passes)
    assert bad <= BAD_PIXELS_MAX, f"{bad:.4f} of pixels deviate"
    assert rm <= 2e-3, rm
    mg, mr = got[fin].mean(0), ref[fin].mean(0)
    assert np.all(np.abs(mg - mr) <= 5e-3 * np.abs(mr) + 1e-12)
    # work counters: the backend traced (almost) exactly the paths the oracle traced
    s = ctx.stats()
    assert s["pixel_samples"] == cn["pixel_samples"] == w * h * passes
    for k, tol in (("eye_vertices", 2e-3), ("light_vertices", 2e-3), ("connections", 5e-3), ("splats", 5e-3)):
        assert abs(s[k] - cn[k]) <= tol * cn[k] + 2, (k, s[k], cn[k])
    # the wavefront pipelines decide Russian roulette before they trace the next ray (same draw, same
    # outcome), so they trace fewer closest-hit rays than the reference order does, never more
    k = "closest_hit_rays"
    if pipeline == 0:
        assert 0.8 * cn[k] < s[k] <= cn[k] * (1 + 2e-3) + 2, (k, s[k], cn[k])
    else:
        assert abs(s[k] - cn[k]) <= 2e-3 * cn[k] + 2, (k, s[k], cn[k])
    ctx.close()


@pytest.mark.parametrize("name", ["C2", "mesh6k"])
def test_pair_queue_overflow_gives_the_same_image(oracle_mod, name, monkeypatch):
    """The pair queue holds 24 connections per pixel-sample of a batch; pixel-samples that no longer fit are
    finished by k_pair_overflow with the reference's loop structure. With room for 2 per pixel-sample most of the
    batch takes that path: the image and the counters must still match the oracle path by path."""
    w = h = 48
    passes = 3
    monkeypatch.setenv("UBDPT_PAIR_QUEUE_PER_PATH", "2")
    scene, ctx, orc = make_pair(oracle_mod, name, w=w, h=h, stats=1)
    monkeypatch.delenv("UBDPT_PAIR_QUEUE_PER_PATH")
    st, done = ctx.render_passes(0, passes, SEED)
    assert st == capi.SUCCESS and done == passes
    got = ctx.read_sum()
    s = ctx.stats()
    ctx.close()
    ref, cn = orc.render_passes(0, passes, SEED, num_threads=4, counters=True)
    bad, rm = image_agreement(got, ref, passes)
    assert bad <= BAD_PIXELS_MAX and rm <= 2e-3, (bad, rm)
    for k, tol in (("connections", 5e-3), ("splats", 5e-3)):
        assert abs(s[k] - cn[k]) <= tol * cn[k] + 2, (k, s[k], cn[k])
    # more connections than the shrunken queue can hold were evaluated: the overflow path did run
    assert s["connections"] > 2 * w * h * passes


@pytest.mark.parametrize("n_spheres", [6, 20, 30, 70])
def test_image_parity_across_scene_sizes_that_change_the_memory_placement(oracle_mod, n_spheres):
    """The kernels keep small scenes in LDS: tree, primitives and objects up to 48 primitives / 16 objects,
    the object table alone up to 64 objects (shading), everything in HBM beyond. 12, 26 / 36 and 76 objects
    select the three placements; the image must match the oracle path by path in each."""
    w = h = 40
    passes = 2
    scene = scenes.many_spheres_scene(n_spheres, seed=n_spheres)
    scene.desc.camera.ratio = 1.0
    params = scenes.make_params(w, h)
    opt = capi.Options()
    opt.collect_stats = 1
    with capi.Context(scene, params, opt) as ctx:
        st, done = ctx.render_passes(0, passes, SEED)
        assert st == capi.SUCCESS and done == passes
        got = ctx.read_sum()
        s = ctx.stats()
        rays = ray_batch(1 << 14, seed=7)
        hits = ctx.intersect_batch(rays)
    orc = oracle_mod.Oracle(scene, params, 32)
    ref, cn = orc.render_passes(0, passes, SEED, num_threads=4, counters=True)
    fin = np.isfinite(ref).all(axis=2) & np.isfinite(got).all(axis=2)
    bad, rm = image_agreement(np.where(fin[..., None], got, 0), np.where(fin[..., None], ref, 0), passes)
    assert bad <= BAD_PIXELS_MAX and rm <= 2e-3, (bad, rm)
    for k in ("eye_vertices", "light_vertices"):
        assert abs(s[k] - cn[k]) <= 2e-3 * cn[k] + 2, (k, s[k], cn[k])
    rh = orc.intersect_batch(rays)
    m = rh["hit"] == 1
    assert np.array_equal(hits["hit"], rh["hit"]) and np.array_equal(hits["object"][m], rh["object"][m])
    assert np.array_equal(hits["d"][m].view(np.uint64), rh["d"][m].view(np.uint64))


@pytest.mark.parametrize("pipeline", [0, 1])
@pytest.mark.parametrize("case", ["roulette_from_first_bounce", "strata_and_wide_lens", "non_square_tiny", "one_pixel"])
def test_image_parity_for_other_render_parameters(oracle_mod, case, pipeline):
    """Render parameters away from the GUI defaults: Russian roulette from the first bounce (min_depth 1; the
    wavefront pipeline decides it before tracing), stratified pixel / lens samples with a wide lens and a near
    focus plane, images that are not square and images of one pixel."""
    w, h, passes, kw = {
        "roulette_from_first_bounce": (40, 40, 3, dict(min_depth=1)),
        "strata_and_wide_lens": (36, 36, 8, dict(pixel_subdiv=2, lens_subdiv=2, lens=0.05, focus=2.5)),
        "non_square_tiny": (7, 3, 5, dict()),
        "one_pixel": (1, 1, 9, dict(min_depth=2)),
    }[case]
    scene = scenes.tiny_scene() if case != "strata_and_wide_lens" else scenes.cornell_c2()
    scene.desc.camera.ratio = w / h
    params = scenes.make_params(w, h, **kw)
    opt = capi.Options()
    opt.collect_stats = 1
    opt.pipeline = pipeline
    with capi.Context(scene, params, opt) as ctx:
        st, done = ctx.render_passes(0, passes, SEED)
        assert st == capi.SUCCESS and done == passes
        got = ctx.read_sum()
        s = ctx.stats()
    orc = oracle_mod.Oracle(scene, params, 32)
    ref, cn = orc.render_passes(0, passes, SEED, num_threads=2, counters=True)
    fin = np.isfinite(ref).all(axis=2) & np.isfinite(got).all(axis=2)
    assert np.array_equal(np.isfinite(ref), np.isfinite(got))
    bad, rm = image_agreement(np.where(fin[..., None], got, 0), np.where(fin[..., None], ref, 0), passes)
    if w * h >= 400:
        assert bad <= BAD_PIXELS_MAX and rm <= 2e-3, (bad, rm)
    else:  # a handful of pixels: every one of them has to agree
        assert bad == 0.0, (got, ref)
    for k in ("eye_vertices", "light_vertices", "connections", "splats"):
        assert abs(s[k] - cn[k]) <= 5e-3 * cn[k] + 2, (k, s[k], cn[k])
    k = "closest_hit_rays"
    assert s[k] <= cn[k] * (1 + 2e-3) + 2 and (pipeline == 0 or abs(s[k] - cn[k]) <= 2e-3 * cn[k] + 2), (k, s[k], cn[k])
    if case == "roulette_from_first_bounce" and pipeline == 0:
        assert s[k] < 0.97 * cn[k]  # the rays of the subpaths that lost the roulette were not traced


def test_scene_without_emitters_and_empty_scene_render_black(oracle_mod):
    """No emitter: light subpaths do not exist and nothing is ever added; no object at all: every ray misses."""
    b = scenes.SceneBuilder(scenes.make_camera((0, 0, -3.4), (0, 0, 0), 0.69, 1.0))
    grey = b.color((0.7, 0.7, 0.7))
    b.sphere((0, 0, 0), 0.5, capi.MAT_LATEX_PAINT, grey)
    b.mesh(scenes.quad((-1, -1, -1), (1, -1, -1), (1, -1, 1), (-1, -1, 1), (0, 1, 0), (1, 0, 0)), scenes.identity(),
           capi.MAT_LATEX_PAINT, grey)
    dark = b.build()
    empty = scenes.SceneBuilder(scenes.make_camera((0, 0, -3.4), (0, 0, 0), 0.69, 1.0)).build()
    params = scenes.make_params(16, 16)
    for scene in (dark, empty):
        for pipeline in (0, 1):
            opt = capi.Options()
            opt.collect_stats = 1
            opt.pipeline = pipeline
            with capi.Context(scene, params, opt) as ctx:
                st, done = ctx.render_passes(0, 2, SEED)
                assert st == capi.SUCCESS and done == 2
                assert not ctx.read_sum().any()
                s = ctx.stats()
                assert s["light_vertices"] == 0 and s["shadow_rays"] == 0
        ref = oracle_mod.Oracle(scene, params, 32).render_passes(0, 2, SEED)
        assert not np.asarray(ref).any()


def test_pass_partition_is_additive(oracle_mod):
    """Passes {0..5} in one call == passes {0,2,4} + {1,3,5} with stride 2 (the multi-GPU partition),
    and a resumed render (upload_sum) continues the same sum."""
    scene, ctx, orc = make_pair(oracle_mod, "C2", w=32, h=32)
    ctx.render_passes(0, 6, SEED)
    full = ctx.read_sum()
    ctx.upload_sum(None)
    ctx.render_passes(0, 3, SEED, pass_stride=2)
    even = ctx.read_sum()
    ctx.upload_sum(None)
    ctx.render_passes(1, 3, SEED, pass_stride=2)
    odd = ctx.read_sum()
    assert np.allclose(even + odd, full, rtol=1e-12, atol=1e-12)
    # resume: upload the even half, render the odd passes on top
    ctx.upload_sum(even)
    ctx.render_passes(1, 3, SEED, pass_stride=2)
    assert np.allclose(ctx.read_sum(), full, rtol=1e-12, atol=1e-12)
    # add_into accumulates into the caller's buffer (the adapter's use)
    acc = np.ones_like(full)
    ctx.read_sum(acc, add_into=True)
    assert np.allclose(acc - 1.0, full, rtol=1e-12, atol=1e-9)
    ctx.close()


@pytest.mark.parametrize("pipeline", [0, 1])
def test_separate_passes_equal_the_passes_rendered_alone(pipeline):
    """ubdpt_render_passes_separate renders a window of passes as one batch but keeps their sums apart: every
    per-pass buffer must equal that pass rendered alone, their total the ordinary render of the window, and the
    device sum proper must be left untouched."""
    scene = SCENES["C2"]()
    scene.desc.camera.ratio = 1.0
    params = scenes.make_params(40, 40)
    opt = capi.Options()
    opt.pipeline = pipeline
    opt.pass_slots = 5
    with capi.Context(scene, params, opt) as ctx:
        ctx.render_passes(100, 2, SEED)
        resident = ctx.read_sum()
        st, done = ctx.render_passes_separate(7, 5, SEED)
        assert st == capi.SUCCESS and done == 5
        slots = [ctx.read_sum_slot(k) for k in range(5)]
        assert np.array_equal(ctx.read_sum(), resident)
        # a second, shorter window: its buffers start from zero again
        ctx.render_passes_separate(3, 2, SEED)
        again = [ctx.read_sum_slot(k) for k in range(2)]
        with pytest.raises(Exception):
            ctx.render_passes_separate(0, 6, SEED)
        ctx.upload_sum(None)
        alone = []
        for p in (7, 8, 9, 10, 11, 3, 4):
            ctx.upload_sum(None)
            ctx.render_passes(p, 1, SEED)
            alone.append(ctx.read_sum())
        ctx.upload_sum(None)
        ctx.render_passes(7, 5, SEED)
        window = ctx.read_sum()
    for k in range(5):
        assert slots[k].sum() > 0 and np.allclose(slots[k], alone[k], rtol=1e-12, atol=1e-12), k
    assert np.allclose(again[0], alone[5], rtol=1e-12, atol=1e-12) and np.allclose(again[1], alone[6], rtol=1e-12, atol=1e-12)
    assert np.allclose(sum(slots), window, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("size", [(40, 40), (700, 520)])
def test_pass_added_onto_another_buffer_equals_copy_then_add(size):
    """ubdpt_read_sum_slot_onto delivers base + pass: the copy of the finished sum and the += of one pass that
    UEngine::renderPass and the renderer do in two sweeps (uengine/uengine.cpp:153-164). Bitwise equal to
    numpy's base + pass for one and for several host threads (the larger image is past the size from which the
    add is split over threads), in place, and with no base at all."""
    w, h = size
    scene = SCENES["C2"]()
    scene.desc.camera.ratio = float(w) / float(h)
    params = scenes.make_params(w, h)
    opt = capi.Options()
    opt.pass_slots = 2
    rng = np.random.default_rng(5)
    base = rng.uniform(0.0, 3.0, (h, w, 3))
    with capi.Context(scene, params, opt) as ctx:
        st, done = ctx.render_passes_separate(4, 2, SEED)
        assert st == capi.SUCCESS and done == 2
        for slot in (0, 1):
            one = ctx.read_sum_slot(slot)
            assert one.sum() > 0
            want = base + one
            for threads in (1, 5, 16, 1000):
                out = np.full((h, w, 3), -7.0)
                ctx.read_sum_slot_onto(slot, base, out, threads)
                assert np.array_equal(out, want), (slot, threads)
            inplace = base.copy()
            ctx.read_sum_slot_onto(slot, inplace, inplace, 4)
            assert np.array_equal(inplace, want)
            plain = np.full((h, w, 3), -7.0)
            ctx.read_sum_slot_onto(slot, None, plain, 4)
            assert np.array_equal(plain, one)
        with pytest.raises(Exception):
            ctx.read_sum_slot_onto(2, base, np.zeros((h, w, 3)), 1)


def test_enqueued_windows_in_two_halves_equal_the_waited_form():
    """ubdpt_enqueue_passes_separate fills one half of the per-pass buffers while the other half is read: every
    buffer must equal the same pass rendered by the waited form, also when a half is re-used for a third window
    and when buffers whose passes were never read are enqueued again."""
    scene = SCENES["C2"]()
    scene.desc.camera.ratio = 1.0
    params = scenes.make_params(48, 48)
    opt = capi.Options()
    opt.pass_slots = 6
    with capi.Context(scene, params, opt) as ctx:
        ref = {}
        for first in (0, 3, 6, 20):
            ctx.render_passes_separate(first, 3, SEED)
            for k in range(3):
                ref[first + k] = ctx.read_sum_slot(k)
        ctx.render_passes(50, 1, SEED)
        resident = ctx.read_sum()
        got = {}
        st, done = ctx.enqueue_passes_separate(0, 3, SEED, 0)
        assert st == capi.SUCCESS and done == 3
        ctx.enqueue_passes_separate(3, 3, SEED, 3)
        for k in range(3):
            got[k] = ctx.read_sum_slot(k)            # waits for the first window only
        ctx.enqueue_passes_separate(6, 3, SEED, 0)   # first half again, while the second is still unread
        for k in range(3):
            got[3 + k] = ctx.read_sum_slot(3 + k)
        ctx.enqueue_passes_separate(20, 3, SEED, 3)  # a jump; then the never-read first half is overwritten
        ctx.enqueue_passes_separate(6, 3, SEED, 0)
        for k in range(3):
            got[20 + k] = ctx.read_sum_slot(3 + k)
            got[6 + k] = ctx.read_sum_slot(k)
        with pytest.raises(Exception):
            ctx.enqueue_passes_separate(0, 3, SEED, 4)   # 4 + 3 > 6 buffers
        ctx.synchronize()
        assert np.array_equal(ctx.read_sum(), resident)  # the device sum proper is left alone
    assert sorted(got) == sorted(ref)
    for p in ref:
        assert ref[p].sum() > 0 and np.allclose(got[p], ref[p], rtol=1e-12, atol=1e-12), p


def test_full_size_headline_workload_properties(oracle_mod):
    """The benchmark configuration at its full size (C2, 512 x 512): too large for a path-by-path comparison in a
    test, so size-independent properties are checked. 64 passes in one call equal the even plus the odd passes
    (the multi-GPU partition), every pixel is finite and nearly all received light, and the mean radiance agrees with the
    oracle's at a size the oracle finishes in a second (the image mean does not depend on the resolution)."""
    scene, params, _ = scenes.make_config("C2")
    with capi.Context(scene, params) as ctx:
        ctx.render_passes(0, 64, SEED)
        full = ctx.read_sum()
        ctx.upload_sum(None)
        ctx.render_passes(0, 32, SEED, pass_stride=2)
        ctx.render_passes(1, 32, SEED, pass_stride=2)
        halves = ctx.read_sum()
    assert np.isfinite(full).all() and (full >= 0).all() and (full.sum(axis=2) > 0).mean() > 0.95
    assert np.allclose(full, halves, rtol=1e-11, atol=1e-11)
    small = scenes.make_params(96, 96)
    sc2 = scenes.cornell_c2()
    sc2.desc.camera.ratio = 1.0
    ref = oracle_mod.Oracle(sc2, small, 32).render_passes(0, 32, SEED + 1, num_threads=4)
    m_gpu, m_ref = full.reshape(-1, 3).mean(0) / 64, ref.reshape(-1, 3).mean(0) / 32
    assert np.all(np.abs(m_gpu - m_ref) <= 0.02 * m_ref), (m_gpu, m_ref)


def test_passes_in_flight_does_not_change_the_result(oracle_mod):
    a = make_pair(oracle_mod, "C1", w=32, h=32, pif=1)[1]
    b = make_pair(oracle_mod, "C1", w=32, h=32, pif=5)[1]
    a.render_passes(0, 7, SEED)
    b.render_passes(0, 7, SEED)
    assert np.allclose(a.read_sum(), b.read_sum(), rtol=1e-12, atol=1e-12)
    a.close(); b.close()


def test_passes_that_do_not_fill_a_batch_wait_for_the_next_call():
    """An enqueue call leaves the passes that do not fill a batch to the next call when that call continues where it
    ended, and launches them itself otherwise (another seed, a gap, a stride, a waited call, a read): whatever the
    calls are cut into, every pass is rendered exactly once and reported by the call that took it."""
    scene = SCENES["C2"]()
    scene.desc.camera.ratio = 1.0
    params = scenes.make_params(40, 40)
    npix = 40 * 40
    opt = capi.Options()
    opt.passes_in_flight = 3
    with capi.Context(scene, params, opt) as ctx:
        ctx.render_passes(0, 25, SEED)
        ref = ctx.read_sum()
        ctx.upload_sum(None)
        # 7 = 2 batches + 1 held; 4 (+1) = 1 batch + 2 held; 2 (+2) = 1 batch + 1 held; a gap: the held pass runs alone;
        # 8 = 2 batches + 2 held; a waited call takes them along: (2 +) 3 = 1 batch + 2 more passes
        done = [ctx.enqueue_passes(0, 7, SEED)[1], ctx.enqueue_passes(7, 4, SEED)[1], ctx.enqueue_passes(11, 2, SEED)[1]]
        enq, fin = ctx.query_progress()
        assert done == [7, 4, 2] and enq - 25 * npix == 13 * npix and fin <= enq
        done.append(ctx.enqueue_passes(14, 8, SEED)[1])
        st, d = ctx.render_passes(22, 3, SEED)
        assert done[-1] == 8 and st == capi.SUCCESS and d == 3
        ctx.render_passes(13, 1, SEED)            # the pass the gap left out
        assert ctx.query_progress() == (50 * npix, 50 * npix)
        got = ctx.read_sum()
        assert ctx.stats()["passes"] == 50
        assert np.allclose(got, ref, rtol=1e-12, atol=1e-12) and ref.sum() > 0
        # passes held when another seed, a stride or a read comes: launched before it
        ctx.upload_sum(None)
        ctx.enqueue_passes(0, 4, SEED)            # 1 batch + 1 held
        ctx.enqueue_passes(4, 5, SEED + 1)        # another seed: the held pass goes first; 1 batch + 2 held
        ctx.enqueue_passes(9, 4, SEED + 1, pass_stride=2)   # another stride
        mixed = ctx.read_sum()                    # waits for everything, held passes included
        ctx.upload_sum(None)
        ctx.render_passes(0, 4, SEED); ctx.render_passes(4, 5, SEED + 1); ctx.render_passes(9, 4, SEED + 1, pass_stride=2)
        assert np.allclose(mixed, ctx.read_sum(), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("pipeline", [0, 1])
def test_sub_pass_batches_and_many_streams_do_not_change_the_result(pipeline):
    """Batches of a few thousand pixels of one pass (never across a pass boundary), spread over 5 streams,
    against whole-pass batches on one stream; also through the asynchronous entry points."""
    scene = SCENES["C2"]()
    scene.desc.camera.ratio = 1.0
    params = scenes.make_params(80, 80)
    imgs = []
    for batch, streams in ((0, 1), (5000, 5)):
        opt = capi.Options()
        opt.pipeline = pipeline
        opt.batch_pixel_samples = batch
        opt.streams = streams
        with capi.Context(scene, params, opt) as ctx:
            ctx.enqueue_passes(0, 2, SEED)
            ctx.enqueue_passes(2, 3, SEED)
            ctx.synchronize()
            imgs.append(ctx.read_sum())
            assert ctx.stats()["passes"] == 5
    assert imgs[0].sum() > 0 and np.allclose(imgs[0], imgs[1], rtol=1e-12, atol=1e-12)


def test_path_cap_matches_oracle(oracle_mod):
    """With a small vertex cap both sides truncate the same subpaths."""
    scene, ctx, orc = make_pair(oracle_mod, "C2", w=32, h=32, maxv=4, stats=1)
    ctx.render_passes(0, 4, SEED)
    got = ctx.read_sum()
    ref, cn = orc.render_passes(0, 4, SEED, counters=True)
    bad, rm = image_agreement(got, ref, 4)
    assert bad <= BAD_PIXELS_MAX and rm <= 2e-3
    s = ctx.stats()
    assert cn["capped_subpaths"] > 0
    assert abs(s["capped_subpaths"] - cn["capped_subpaths"]) <= 0.01 * cn["capped_subpaths"] + 2
    ctx.close()


def test_stop_request_is_honoured(oracle_mod):
    import threading
    import time
    scene, ctx, orc = make_pair(oracle_mod, "C1", w=256, h=256, pif=1)
    t = threading.Timer(0.05, ctx.request_stop)
    t.start()
    st, done = ctx.render_passes(0, 100000, SEED)
    t.join()
    assert st == capi.STOPPED and 0 <= done < 100000
    # the context stays usable and a new render call resets the stop flag (ubdptrenderer.cpp:37)
    st, done2 = ctx.render_passes(done, 1, SEED)
    assert st == capi.SUCCESS and done2 == 1
    ctx.close()


@pytest.mark.parametrize("n_tris,n_rays", [(100_000, 1 << 14), (2_000_000, 1 << 11)])
def test_large_mesh_lbvh_equals_linear_scan(n_tris, n_rays):
    """At the sizes of the mesh configurations the CPU oracle's O(#triangles) scan is too slow for a test,
    so the LBVH is checked against the device's own linear scan, which runs the same fp64 leaf tests in
    the reference's order (and is itself checked against the oracle on the small scenes above)."""
    scene = scenes.mesh_scene(n_tris, seed=1, ratio=1.0)
    params = scenes.make_params(64, 64)
    rays, segs = ray_batch(n_rays, seed=42), segment_batch(n_rays, seed=43)
    res = []
    for accel in (0, 1):
        opt = capi.Options()
        opt.accel = accel
        with capi.Context(scene, params, opt) as ctx:
            res.append((ctx.intersect_batch(rays), ctx.visibility_batch(segs)))
    (h0, v0), (h1, v1) = res
    assert np.array_equal(h0["hit"], h1["hit"]) and np.array_equal(h0["object"], h1["object"])
    assert np.array_equal(h0["face"], h1["face"])
    m = h1["hit"] == 1
    assert m.sum() > n_rays // 2
    for f in ("d", "pos", "Ng", "Ns", "Ts", "Bs", "tex_u", "tex_v"):
        assert np.array_equal(h0[f][m].view(np.uint64), h1[f][m].view(np.uint64)), f
    assert np.array_equal(v0, v1) and 0 < v1.sum() < n_rays


@pytest.mark.parametrize("shift", [(0.0, 0.0, 0.0), (300.0, -200.0, 500.0), (5000.0, 3000.0, -4000.0)])
def test_tree_never_loses_a_hit_on_edges_corners_and_far_from_the_origin(shift):
    """The tree is walked in fp32 through quantised boxes; what keeps it exact is that a box is never rejected for a
    ray the fp64 primitive test accepts. The rays that probe this are the ones through the rim of a box: aimed at
    corners and edge points of triangles, from near and far, with axis-parallel directions among them, and the
    same with the whole scene moved far from the origin, where fp32 coordinates are coarsest. Hit identity and
    every hit field must equal the device's linear scan bit for bit; visibility likewise."""
    n_tris, n_rays = 20_000, 1 << 15
    scene = scenes.mesh_scene(n_tris, seed=3, ratio=1.0)
    sh = np.asarray(shift)
    for i in range(scene.n_objects):
        for k in range(3):
            scene.objects[i].W[12 + k] += sh[k]
    cam = scene.desc.camera
    for k in range(3):
        cam.pos[k] += sh[k]; cam.look_at[k] += sh[k]
    scene.desc.camera = cam
    # world-space corners of random faces
    rng = np.random.default_rng(11)
    face_obj = np.zeros(scene.num_faces, np.int64)
    Ws = []
    for i in range(scene.n_objects):
        o = scene.objects[i]
        Ws.append(np.array(list(o.W)).reshape(4, 4).T)
        if o.kind == capi.MODEL_MESH:
            face_obj[o.first_face:o.first_face + o.num_faces] = i
    f = rng.integers(0, scene.num_faces, n_rays)
    P = scene.vertices["pos"].reshape(-1, 3, 3)[f].astype(np.float64)           # [n][corner][xyz], object-local
    Wf = np.stack([Ws[i] for i in face_obj[f]])
    Pw = np.einsum("nij,ncj->nci", Wf[:, :3, :3], P) + Wf[:, None, :3, 3]
    bary = rng.dirichlet((1, 1, 1), n_rays)
    kind = rng.integers(0, 3, n_rays)
    bary[kind == 0] = np.eye(3)[rng.integers(0, 3, (kind == 0).sum())]            # a corner
    e = kind == 1                                                                  # a point of an edge
    bary[e, rng.integers(0, 3, e.sum())] = 0.0
    bary[e] /= bary[e].sum(axis=1, keepdims=True)
    target = np.einsum("nc,nci->ni", bary, Pw)
    d = rng.normal(size=(n_rays, 3))
    axis = rng.integers(0, 4, n_rays)                                              # a quarter of the rays run along an axis
    for a in range(3):
        d[axis == a] = np.eye(3)[a] * rng.choice([-1.0, 1.0], (axis == a).sum())[:, None]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    dist = rng.choice([1e-3, 0.05, 0.7, 3.0, 40.0], n_rays)
    rays = np.zeros(n_rays, capi.RAY_DTYPE)
    rays["origin"] = target - d * dist[:, None]
    rays["dir"] = d
    segs = np.zeros(n_rays, capi.SEGMENT_DTYPE)
    segs["p0"] = rays["origin"]
    segs["p1"] = target + d * rng.choice([-1e-4, 1e-4, 0.3], n_rays)[:, None]
    res = []
    for accel in (0, 1):
        opt = capi.Options()
        opt.accel = accel
        with capi.Context(scene, scenes.make_params(32, 32), opt) as ctx:
            res.append((ctx.intersect_batch(rays), ctx.visibility_batch(segs)))
    (h0, v0), (h1, v1) = res
    assert np.array_equal(h0["hit"], h1["hit"]) and np.array_equal(h0["object"], h1["object"])
    assert np.array_equal(h0["face"], h1["face"])
    m = h1["hit"] == 1
    assert m.sum() > n_rays // 2
    for fld in ("d", "pos", "Ng", "Ns", "Ts", "Bs", "tex_u", "tex_v"):
        assert np.array_equal(h0[fld][m].view(np.uint64), h1[fld][m].view(np.uint64)), fld
    assert np.array_equal(v0, v1) and 0 < v1.sum() < n_rays


@pytest.mark.parametrize("name", ["C1", "C2"])
def test_relmse_gate_against_independent_reference_seeds(oracle_mod, name):
    """The statistical gate for images: at equal spp the backend's image must be as close to an
    independently seeded CPU image as two CPU images are to each other:
    relMSE(GPU(seed 3), REF(seed 1)) <= 1.5 * relMSE(REF(seed 2), REF(seed 1)), channel means within 1 %."""
    n = 32
    scene, ctx, orc = make_pair(oracle_mod, name, w=64, h=64)
    ref1 = orc.render_passes(0, n, 1, num_threads=8) / n
    ref2 = orc.render_passes(0, n, 2, num_threads=8) / n
    ctx.render_passes(0, n, 3)
    gpu = ctx.read_sum() / n
    ctx.close()
    noise = rel_mse(ref2, ref1)
    assert rel_mse(gpu, ref1) <= 1.5 * noise
    m_gpu, m_ref = gpu.reshape(-1, 3).mean(0), 0.5 * (ref1 + ref2).reshape(-1, 3).mean(0)
    assert np.all(np.abs(m_gpu - m_ref) / m_ref < 0.01)


def stacked_triangles_scene(n=32768):
    """n large triangles stacked 2e-5 apart along z: a ray along z meets every box of the tree on its way down,
    so the traversal stack grows by three entries per four-wide level."""
    b = scenes.SceneBuilder(scenes.make_camera((0, 0, -3.4), (0, 0, 0), 0.69, 1.0))
    grey = b.color((0.7, 0.7, 0.7))
    z = (np.arange(n) * 2e-5)[:, None]
    tri = np.zeros((n, 3), capi.MESH_VERTEX_DTYPE)
    tri["pos"][:, 0] = np.concatenate([np.full((n, 1), -0.8), np.full((n, 1), -0.5), z], 1)
    tri["pos"][:, 1] = np.concatenate([np.full((n, 1), 0.8), np.full((n, 1), -0.5), z], 1)
    tri["pos"][:, 2] = np.concatenate([np.full((n, 1), 0.0), np.full((n, 1), 0.9), z], 1)
    tri["normal"] = (0, 0, -1)
    tri["tangent"] = (1, 0, 0)
    b.mesh(tri.reshape(-1), scenes.identity(), capi.MAT_LATEX_PAINT, grey)
    b.mesh(scenes.quad((-.25, .999, -.25), (-.25, .999, .25), (.25, .999, .25), (.25, .999, -.25), (0, -1, 0), (1, 0, 0)),
           scenes.identity(), capi.MAT_LATEX_PAINT, grey, power=(20, 20, 20))
    return b.build()


def test_traversal_stack_overflow_is_reported_not_hidden():
    """A ray whose traversal stack is full would drop a subtree and could miss a hit. With the stack cut to
    6 entries the stacked-triangle scene overflows it: every entry point that traced such a ray must return
    UBDPT_ERROR. With the default stack the same scene renders and the tree agrees with the linear scan."""
    scene = stacked_triangles_scene()
    params = scenes.make_params(32, 32)
    rays = np.zeros(256, capi.RAY_DTYPE)
    rays["origin"] = (0.0, 0.0, -1.0)
    rays["origin"][:, 0] = np.linspace(-0.2, 0.2, 256)
    rays["dir"] = (0.0, 0.0, 1.0)
    segs = np.zeros(256, capi.SEGMENT_DTYPE)
    segs["p0"] = rays["origin"]
    segs["p1"] = rays["origin"] + np.array((0.3, 0.2, 0.99))  # ends in front of the first triangle: nothing blocks
    opt = capi.Options()
    opt.traversal_stack_entries = 6
    with capi.Context(scene, params, opt) as ctx:
        with pytest.raises(capi.UbdptError) as e:
            ctx.intersect_batch(rays)
        assert e.value.status == capi.ERROR
        with pytest.raises(capi.UbdptError) as e:
            ctx.render_passes(0, 1, SEED)
        assert e.value.status == capi.ERROR
        # the failure is reported once; a call that overflows nothing succeeds again on the same context
        assert ctx.visibility_batch(segs).all()
    res = []
    for accel in (0, 1):
        opt = capi.Options()
        opt.accel = accel
        with capi.Context(scene, params, opt) as ctx:
            res.append(ctx.intersect_batch(rays))
            if accel == 0:
                st, done = ctx.render_passes(0, 1, SEED)
                assert st == capi.SUCCESS and done == 1
    assert np.array_equal(res[0]["face"], res[1]["face"]) and (res[0]["face"] == 0).all()
    assert np.array_equal(res[0]["d"].view(np.uint64), res[1]["d"].view(np.uint64))


def test_large_mesh_emitter_is_sampled_like_the_reference_walk(oracle_mod):
    """An emitter mesh of 3000 faces: the reference picks the face by walking the face probabilities
    (uapp/mesh.cpp:206-232, O(#faces) per light subpath), the backend bisects a table of running sums for meshes of
    more than 64 faces. The two can only differ within rounding of a boundary between two faces, so the image
    must match the oracle's path by path like every other scene."""
    b = scenes.SceneBuilder(scenes.make_camera((0, 0, -3.4), (0, 0, 0), 0.69, 1.0))
    scenes._cornell_shell(b)
    lamp = scenes.displaced_sphere(3000, seed=9, amp=0.1)
    b.mesh(lamp, scenes.mesh_transform((0.3, 0.2, 0.1), 0.5, (0, 1, 0), (0.15, 0.1, 0.15)), capi.MAT_LATEX_PAINT,
           b.color((0.8, 0.8, 0.8)), power=(6, 5, 4))
    scene = b.build()
    w = h = 48
    passes = 3
    params = scenes.make_params(w, h)
    opt = capi.Options()
    opt.collect_stats = 1
    with capi.Context(scene, params, opt) as ctx:
        st, done = ctx.render_passes(0, passes, SEED)
        assert st == capi.SUCCESS and done == passes
        got = ctx.read_sum()
        s = ctx.stats()
    ref, cn = oracle_mod.Oracle(scene, params, 32).render_passes(0, passes, SEED, num_threads=8, counters=True)
    bad, rm = image_agreement(got, ref, passes)
    print(f"mesh emitter: {bad * 100:.3f} % of pixels off by > 1e-9")
    assert bad <= BAD_PIXELS_MAX and rm <= 2e-3, (bad, rm)
    for k in ("eye_vertices", "light_vertices", "connections", "splats"):
        assert abs(s[k] - cn[k]) <= 2e-3 * cn[k] + 2, (k, s[k], cn[k])


def _probe_compare(ctx, orc, eye, light, what):
    got, got_s0 = ctx.connection_probe(eye, light)
    ref, ref_s0 = orc.connection_probe(eye, light)
    assert np.array_equal(got["connected"], ref["connected"]), what
    m = ref["connected"] == 1
    worst = 0.0
    if m.any():
        for f in ("c", "w"):
            g, r = got[f][m], ref[f][m]
            assert np.isfinite(r).all() == np.isfinite(g).all(), (what, f)
            fin = np.isfinite(r)
            rel = np.abs(g[fin] - r[fin]) / np.maximum(np.abs(r[fin]), 1e-300)
            worst = max(worst, float(rel.max()) if rel.size else 0.0)
    fin = np.isfinite(ref_s0)
    assert np.array_equal(fin, np.isfinite(got_s0)), what
    rel = np.abs(got_s0[fin] - ref_s0[fin]) / np.maximum(np.abs(ref_s0[fin]), 1e-300)
    rel = rel[np.abs(ref_s0[fin]) > 0]
    assert np.array_equal(got_s0[fin] == 0, ref_s0[fin] == 0), what
    worst = max(worst, float(rel.max()) if rel.size else 0.0)
    return int(m.sum()), int((ref_s0 != 0).any(axis=1).sum()), worst


def test_connection_factor_mis_weight_and_s0_sample_probes(oracle_mod):
    """connectionFactor, weight and s0sample (ubdptrenderer.cpp:156-222, 596-844) of the device code, probed directly
    on subpaths handed in as vertex arrays: the six subpath pairs frozen in tests/golden/paths_C2.npz and the pairs the
    oracle traces for a few hundred pixels of the C2 and the tiny scene (specular chains, s = 1, t = 1, emitter hits).
    The same (s, t) must connect, and c, w and the s = 0 value agree to 1e-12 relative (the device evaluates the
    Lambertian case in a simplified but algebraically identical form)."""
    from util import load_golden
    g = load_golden("paths_C2.npz")
    scene, ctx, orc = make_pair(oracle_mod, "C2", w=64, h=64)
    n_conn = n_s0 = 0
    worst = 0.0
    for k in range(6):
        a, b, wv = _probe_compare(ctx, orc, g[f"p{k}_eye"], g[f"p{k}_light"], f"golden path {k}")
        n_conn += a; n_s0 += b; worst = max(worst, wv)
    ctx.close()
    for name in ("C2", "tiny"):
        scene, ctx, orc = make_pair(oracle_mod, name, w=64, h=64)
        rng = np.random.default_rng(5)
        for _ in range(150):
            px, py, p = int(rng.integers(0, 64)), int(rng.integers(0, 64)), int(rng.integers(0, 8))
            eye, light, I, splats = orc.trace_pixel(p, SEED, px, py)
            if len(light) == 0 or len(eye) == 0:
                continue
            a, b, wv = _probe_compare(ctx, orc, eye, light, f"{name} pixel {px},{py} pass {p}")
            n_conn += a; n_s0 += b; worst = max(worst, wv)
        ctx.close()
    print(f"connection probes: {n_conn} connected pairs, {n_s0} s=0 samples, worst relative deviation {worst:.2e}")
    assert n_conn > 500 and n_s0 > 5
    assert worst <= 1e-12

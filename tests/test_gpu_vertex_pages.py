"""Vertex slots beyond the first `dense_vertex_slots` of a subpath live in pages handed out from a pool
(ubdpt_options.dense_vertex_slots / vertex_pool_pages). The layout must not change a single path: same image as with
every slot stored densely, same agreement with the CPU oracle; and a batch that exhausts the pool must neither truncate
subpaths nor fail: its sums are dropped on the device and it is rendered again in pieces the pool can serve."""
import os
import numpy as np
import pytest

from util import capi, scenes

pytestmark = pytest.mark.gpu

SEED = 0x5EED


def glass_box():
    """Two glass spheres and a mirror wall in the box: subpaths of every length up to the cap."""
    b = scenes.SceneBuilder(scenes.make_camera((0, 0, -3.4), (0, 0, 0), 0.69, 1.0))
    white = b.color((1.0, 1.0, 1.0)); grey = b.color((0.725, 0.725, 0.725))
    scenes._cornell_shell(b, grey=grey)
    b.sphere((-0.4, -0.55, 0.2), 0.42, capi.MAT_DIELECTRIC, white, mat_a=1.5)
    b.sphere((0.45, -0.1, -0.1), 0.4, capi.MAT_DIELECTRIC, white, mat_a=1.33)
    b.mesh(scenes.quad((-1, -1, 0.99), (1, -1, 0.99), (1, 1, 0.99), (-1, 1, 0.99), (0, 0, -1), (1, 0, 0)), scenes.identity(),
           capi.MAT_PERFECT_MIRROR, white)
    return b.build()


def render(scene, params, passes, **kw):
    opt = capi.Options()
    opt.collect_stats = 1
    for k, v in kw.items():
        setattr(opt, k, v)
    with capi.Context(scene, params, opt) as ctx:
        st, done = ctx.render_passes(0, passes, SEED)
        assert st == capi.SUCCESS and done == passes
        return ctx.read_sum(), ctx.stats()


@pytest.mark.parametrize("pipeline", [0, 1])
@pytest.mark.parametrize("name", ["C2", "glass"])
def test_paged_slots_give_the_image_of_the_dense_layout_and_of_the_oracle(name, pipeline):
    """Eight dense slots, the rest paged: on the box with the glass and the mirror sphere 1.6 % of the subpaths go
    beyond slot 8, in a box of two glass spheres and a mirror wall many more and some beyond slots 16 and 24 (three
    page levels). Against 32 dense slots: the same image to the rounding of the sums; against the oracle: path by path."""
    import oracle_py
    scene = scenes.cornell_c2() if name == "C2" else glass_box()
    params = scenes.make_params(64, 64)
    passes = 5
    paged, sp = render(scene, params, passes, dense_vertex_slots=8, pipeline=pipeline)
    dense, sd = render(scene, params, passes, dense_vertex_slots=32, pipeline=pipeline)
    assert dense.sum() > 0 and np.allclose(paged, dense, rtol=1e-12, atol=1e-12)
    for k in ("closest_hit_rays", "shadow_rays", "eye_vertices", "light_vertices", "connections", "splats", "capped_subpaths"):
        assert sp[k] == sd[k], k
    ref, cn = oracle_py.Oracle(scene, params, 32).render_passes(0, passes, SEED, num_threads=8, counters=True)
    deep = (cn["eye_vertices"] + cn["light_vertices"]) / (2.0 * 64 * 64 * passes)
    fin = np.isfinite(ref).all(axis=2) & np.isfinite(paged).all(axis=2)
    g, r = np.where(fin[..., None], paged, 0) / passes, np.where(fin[..., None], ref, 0) / passes
    err = np.abs(g - r) / np.maximum(np.abs(r), 1e-3)
    bad = float(np.any(err > 1e-9, axis=2).mean())
    print(f"{name}, pipeline {pipeline}: {deep:.2f} vertices per subpath, {bad * 100:.3f} % of pixels off by > 1e-9")
    assert bad <= 0.003, bad
    for k in ("eye_vertices", "light_vertices", "connections"):
        assert abs(sp[k] - cn[k]) <= 5e-3 * cn[k] + 2, (k, sp[k], cn[k])


@pytest.fixture
def tiny_pool():
    """A pool of 48 pages per lane and kind of subpath, whatever the batch size (the library never makes it smaller than
    1024 subpaths can fill; UBDPT_POOL_PAGES is its hook for this test)."""
    os.environ["UBDPT_POOL_PAGES"] = "48"
    yield 48
    del os.environ["UBDPT_POOL_PAGES"]


def oracle_agreement(got, scene, params, passes):
    import oracle_py
    ref = oracle_py.Oracle(scene, params, 32).render_passes(0, passes, SEED, num_threads=8)
    fin = np.isfinite(ref).all(axis=2) & np.isfinite(got).all(axis=2)
    g, r = np.where(fin[..., None], got, 0) / passes, np.where(fin[..., None], ref, 0) / passes
    err = np.abs(g - r) / np.maximum(np.abs(r), 1e-3)
    return float(np.any(err > 1e-9, axis=2).mean())


@pytest.mark.parametrize("name,pipeline", [("C2", 0), ("C2", 1), ("glass", 0)])
def test_a_batch_that_exhausts_the_pool_is_rendered_again_in_pieces(tiny_pool, name, pipeline):
    """Four passes of the glass-and-mirror box at 96 x 96 are one batch of 36 864 pixel-samples whose subpaths ask for
    several hundred pages; the pool has 48. The call succeeds, the image is the one of the dense layout (to the rounding
    of the sums) and of the oracle (path by path), the counters are those of the dense run (the work of the dropped
    batches is not counted), and the context reports batches rendered again and a smaller batch from then on. In the box
    of glass spheres the pool is smaller still, so that some of the pieces fail again and are cut once more."""
    scene = scenes.cornell_c2() if name == "C2" else glass_box()
    if name == "glass":   # sixteen pages: pieces of a few hundred pixel-samples, some of which need more than sixteen
        os.environ["UBDPT_POOL_PAGES"] = "16"
    params = scenes.make_params(96, 96)
    passes = 4
    opt = capi.Options()
    opt.collect_stats = 1
    opt.dense_vertex_slots = 8
    opt.pipeline = pipeline
    with capi.Context(scene, params, opt) as ctx:
        before, _, _ = ctx.batch_info()
        st, done = ctx.render_passes(0, passes, SEED)
        assert st == capi.SUCCESS and done == passes
        paged, sp = ctx.read_sum(), ctx.stats()
        again, now = ctx.recovery_info()
        # more passes on the same context: batches are small enough from the start now
        st, done = ctx.render_passes(passes, 2, SEED)
        assert st == capi.SUCCESS and done == 2
        more = ctx.read_sum()
        again2, now2 = ctx.recovery_info()
    print(f"{name}, pipeline {pipeline}: {again} batches rendered again, {before} -> {now} pixel-samples per batch; "
          f"after two more passes {again2}, {now2}")
    assert again >= (1 if name == "C2" else 2) and now < passes * 96 * 96 and now2 <= now
    dense, sd = render(scene, params, passes, dense_vertex_slots=32, pipeline=pipeline)
    assert dense.sum() > 0 and np.allclose(paged, dense, rtol=1e-12, atol=1e-12)
    for k in ("closest_hit_rays", "shadow_rays", "eye_vertices", "light_vertices", "connections", "splats", "capped_subpaths"):
        assert sp[k] == sd[k], k
    assert oracle_agreement(paged, scene, params, passes) <= 0.003
    dense6, _ = render(scene, params, passes + 2, dense_vertex_slots=32, pipeline=pipeline)
    assert np.allclose(more, dense6, rtol=1e-12, atol=1e-12)


def test_exhausted_batches_of_passes_kept_apart_are_complete_when_their_buffers_are_read(tiny_pool):
    """The same through the per-pass buffers (ubdpt_enqueue_passes_separate, what the reference-shaped renderPass is
    served from): a buffer is read after the fence of the call that filled it, and a batch of that call that had to be
    rendered again must be in it by then. Two windows of four passes, the second enqueued while the first is read, then
    the first window's buffers reused."""
    scene = scenes.cornell_c2()
    params = scenes.make_params(96, 96)
    opt = capi.Options()
    opt.dense_vertex_slots = 8
    opt.pass_slots = 8
    with capi.Context(scene, params, opt) as ctx:
        got = {}
        ctx.enqueue_passes_separate(0, 4, SEED, 0)
        ctx.enqueue_passes_separate(4, 4, SEED, 4)
        for k in range(4):
            got[k] = ctx.read_sum_slot(k)
        ctx.enqueue_passes_separate(8, 4, SEED, 0)
        for k in range(4, 8):
            got[k] = ctx.read_sum_slot(k)
        for k in range(4):
            got[8 + k] = ctx.read_sum_slot(k)
        again, now = ctx.recovery_info()
    assert again >= 1
    opt = capi.Options()
    opt.dense_vertex_slots = 32
    with capi.Context(scene, params, opt) as ctx:
        for k in (0, 3, 5, 8, 11):
            ctx.upload_sum(None)
            ctx.render_passes(k, 1, SEED)
            assert np.allclose(got[k], ctx.read_sum(), rtol=1e-12, atol=1e-12), k
    print(f"{again} batches rendered again, {now} pixel-samples per batch at the end")

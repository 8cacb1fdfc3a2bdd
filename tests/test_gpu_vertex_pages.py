"""Vertex slots beyond the first `dense_vertex_slots` of a subpath live in pages handed out from a pool
(ubdpt_options.dense_vertex_slots / vertex_pool_pages). The layout must not change a single path: same image as with
every slot stored densely, same agreement with the CPU oracle; and a pool that is too small must make the call fail
instead of truncating subpaths silently."""
import numpy as np
import pytest

from util import capi, scenes

pytestmark = pytest.mark.gpu

SEED = 0x5EED


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
    if name == "C2":
        scene = scenes.cornell_c2()
    else:
        b = scenes.SceneBuilder(scenes.make_camera((0, 0, -3.4), (0, 0, 0), 0.69, 1.0))
        white = b.color((1.0, 1.0, 1.0)); grey = b.color((0.725, 0.725, 0.725))
        scenes._cornell_shell(b, grey=grey)
        b.sphere((-0.4, -0.55, 0.2), 0.42, capi.MAT_DIELECTRIC, white, mat_a=1.5)
        b.sphere((0.45, -0.1, -0.1), 0.4, capi.MAT_DIELECTRIC, white, mat_a=1.33)
        b.mesh(scenes.quad((-1, -1, 0.99), (1, -1, 0.99), (1, 1, 0.99), (-1, 1, 0.99), (0, 0, -1), (1, 0, 0)), scenes.identity(),
               capi.MAT_PERFECT_MIRROR, white)
        scene = b.build()
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


def test_an_empty_page_pool_fails_the_call():
    """One page per 1024 subpaths on the glass-and-mirror box: the batch needs more, and the call must say so."""
    scene = scenes.cornell_c2()
    params = scenes.make_params(96, 96)
    opt = capi.Options()
    opt.dense_vertex_slots = 8
    opt.vertex_pool_pages = 1
    with capi.Context(scene, params, opt) as ctx:
        with pytest.raises(capi.UbdptError) as e:
            ctx.render_passes(0, 4, SEED)
        assert e.value.status == capi.ERROR
    opt.vertex_pool_pages = 256
    with capi.Context(scene, params, opt) as ctx:
        st, done = ctx.render_passes(0, 4, SEED)
        assert st == capi.SUCCESS and done == 4

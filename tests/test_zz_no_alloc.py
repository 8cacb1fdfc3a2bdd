"""The entry point that allocates nothing (ubdpt_intersect_batch_noalloc) against ubdpt_intersect_batch.

Rays and hits are device buffers of the caller (allocated by the test through the HIP runtime); the kernel's
scratch memory is a static array inside the library. Bar: every field of every hit bit-exact, on two
calls in a row, the second with a smaller input (the array keeps the first call's contents)."""
import numpy as np
import pytest

from util import DeviceBuffer, capi, ray_batch, scenes

pytestmark = pytest.mark.gpu


def _device_call(ctx, rays):
    n = rays.shape[0]
    with DeviceBuffer(rays.nbytes) as d_rays, DeviceBuffer(n * capi.HIT_DTYPE.itemsize, fill=0xAB) as d_hits:
        d_rays.upload(rays)
        ctx.intersect_batch_noalloc(d_rays.ptr, n, d_hits.ptr)
        return d_hits.download(capi.HIT_DTYPE, n)


def _same(a, b):
    for f in capi.HIT_DTYPE.names:
        x, y = np.ascontiguousarray(a[f]), np.ascontiguousarray(b[f])
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), f


@pytest.mark.parametrize("name", ["tiny", "mesh"])
def test_no_alloc_entry_point_matches_the_allocating_one(name):
    scene = scenes.tiny_scene() if name == "tiny" else scenes.mesh_scene(20000, seed=5, ratio=1.0)
    params = scenes.make_params(32, 32)
    with capi.Context(scene, params) as ctx:
        first = ray_batch(100_000, seed=11)       # more rays than lanes: the stride loop runs several times
        second = ray_batch(3_001, seed=12)        # smaller, not a multiple of the workgroup size
        ref1, ref2 = ctx.intersect_batch(first), ctx.intersect_batch(second)
        got1 = _device_call(ctx, first)
        got2 = _device_call(ctx, second)
        assert ref1["hit"].sum() > 50_000
        _same(got1, ref1)
        _same(got2, ref2)
        # sizes beyond the stated maximum are refused, an empty batch is fine
        with DeviceBuffer(256) as dummy:
            with pytest.raises(capi.UbdptError) as e:
                ctx.intersect_batch_noalloc(dummy.ptr, capi.NOALLOC_MAX_RAYS + 1, dummy.ptr)
            assert e.value.status == capi.INVALID_ARG
            ctx.intersect_batch_noalloc(dummy.ptr, 0, dummy.ptr)
            # passes still enqueued: the call refuses to wait for them itself
            ctx.enqueue_passes(0, 1, 0x5EED)
            with pytest.raises(capi.UbdptError) as e:
                ctx.intersect_batch_noalloc(dummy.ptr, 1, dummy.ptr)
            assert e.value.status == capi.ERROR
            ctx.synchronize()
        _same(_device_call(ctx, second), ref2)


def test_no_alloc_entry_point_reports_stack_overflow():
    import importlib
    t = importlib.import_module("test_gpu_parity")
    scene = t.stacked_triangles_scene()
    rays = np.zeros(64, capi.RAY_DTYPE)
    rays["origin"] = (0.0, 0.0, -1.0)
    rays["dir"] = (0.0, 0.0, 1.0)
    opt = capi.Options()
    opt.traversal_stack_entries = 6
    with capi.Context(scene, scenes.make_params(8, 8), opt) as ctx:
        with pytest.raises(capi.UbdptError) as e:
            _device_call(ctx, rays)
        assert e.value.status == capi.ERROR
    with capi.Context(scene, scenes.make_params(8, 8)) as ctx:
        _same(_device_call(ctx, rays), ctx.intersect_batch(rays))

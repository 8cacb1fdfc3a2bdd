"""Batch buffers are sized from a snapshot of the free device memory (four fifths of it). When the snapshot is too
optimistic -- another context took the memory in between, as when two ranks start on one GPU at the same moment --
ubdpt_create shrinks the batches (half the passes, one lane, parts of a pass) and tries again. The image must not
depend on which size it ended up with."""
import threading

import numpy as np
import pytest

from util import capi, scenes

pytestmark = pytest.mark.gpu

SEED = 0x5EED


def test_create_shrinks_its_batches_when_the_memory_snapshot_was_too_optimistic(monkeypatch):
    """UBDPT_ASSUME_FREE_BYTES stands in for a stale snapshot: 64 TB "free". With 255 vertex slots per subpath a pass
    of 1280 x 720 needs 90 GB per lane, the four passes per batch the sizing then picks cannot be allocated, and the
    retry path has to come down to what fits. Same image as a context that was sized from the real figure."""
    scene = scenes.cornell_c2()
    params = scenes.make_params(1280, 720)
    scene.desc.camera.ratio = 1280 / 720
    opt = capi.Options()
    opt.max_path_vertices = 255
    with capi.Context(scene, params, opt) as ctx:
        st, done = ctx.render_passes(0, 2, SEED)
        assert st == capi.SUCCESS and done == 2
        ref = ctx.read_sum()
    monkeypatch.setenv("UBDPT_ASSUME_FREE_BYTES", str(64 << 40))
    with capi.Context(scene, params, opt) as ctx:
        st, done = ctx.render_passes(0, 2, SEED)
        assert st == capi.SUCCESS and done == 2
        got = ctx.read_sum()
    assert ref.sum() > 0 and np.allclose(got, ref, rtol=1e-12, atol=1e-12)


def test_two_contexts_created_at_the_same_moment_on_one_device():
    """Two threads call ubdpt_create for a 1080p rendering on device 0 at once: both read the same free-memory figure
    and both want four fifths of it. Both must come back with a working context and render the same image."""
    scene = scenes.cornell_c2()
    params = scenes.make_params(1920, 1080)
    scene.desc.camera.ratio = 16 / 9
    out = [None, None]
    err = []
    gate = threading.Barrier(2)

    def work(k):
        try:
            gate.wait()
            with capi.Context(scene, params) as ctx:
                gate.wait()   # both contexts exist at the same time
                st, done = ctx.render_passes(0, 2, SEED)
                assert st == capi.SUCCESS and done == 2
                out[k] = ctx.read_sum()
        except Exception as e:  # noqa: BLE001 - reported below
            err.append(e)
            gate.abort()

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not err, err
    assert out[0].sum() > 0 and np.allclose(out[0], out[1], rtol=1e-12, atol=1e-12)

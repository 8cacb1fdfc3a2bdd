"""The multi-GPU path on CPU: world_size-2 (and 3) `gloo` process groups run the same partition and
reduce code bench.py runs over RCCL, with the CPU oracle standing in for the renderer of each rank.
The reduced sum must equal the single-process render of the same passes."""
import os
import socket
import sys

import numpy as np
import pytest

from util import ROOT, pkg, scenes

multigpu = pkg.multigpu


def test_partition_covers_every_pass_exactly_once():
    for first, n, world in [(0, 16, 1), (0, 16, 2), (5, 17, 4), (3, 2, 8), (0, 0, 2), (100, 7, 3)]:
        seen = []
        for r in range(world):
            f, c, s = multigpu.partition_passes(first, n, r, world)
            assert s == world
            seen += [f + k * s for k in range(c)]
        assert sorted(seen) == list(range(first, first + n))
    with pytest.raises(ValueError):
        multigpu.partition_passes(0, 4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path, passes):
    import torch
    import torch.distributed as dist
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_py
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc, params, _ = scenes.make_config("C2")
    params.img_res_x = params.img_res_y = 24
    orc = oracle_py.Oracle(sc, params)
    first, count, stride = multigpu.partition_passes(0, passes, rank, world)
    img = orc.render_passes(first, count, 0x5EED, pass_stride=stride) if count else np.zeros((24, 24, 3))
    t = torch.from_numpy(img.copy())
    dist.barrier()
    multigpu.reduce_sum(t, dist)
    if rank == 0:
        np.save(out_path, t.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_ranks_reduce_to_the_full_render(tmp_path, world):
    import torch.multiprocessing as mp
    import oracle_py
    passes = 7
    out = str(tmp_path / "sum.npy")
    mp.start_processes(_worker, args=(world, _free_port(), out, passes), nprocs=world, join=True, start_method="spawn")
    got = np.load(out)
    sc, params, _ = scenes.make_config("C2")
    params.img_res_x = params.img_res_y = 24
    ref = oracle_py.Oracle(sc, params).render_passes(0, passes, 0x5EED)
    assert np.allclose(got, ref, rtol=1e-12, atol=1e-13)
    assert got.sum() > 0

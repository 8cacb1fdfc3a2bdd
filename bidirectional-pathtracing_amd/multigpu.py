"""Pass partitioning over GPUs and the final reduction of the per-GPU sum buffers.

Passes are independent Monte-Carlo samples that only meet in `+=` on the sum buffer
(uengine/ubdptrenderer.cpp:144,153), and the pass index only selects the stratum (:95-96). So rank
r of G renders passes r, r+G, r+2G, ... of the requested range for the whole image (pixel tiles
would not work: t=1 splats land in arbitrary pixels), with no communication while rendering, and
one reduce(sum) combines the buffers at the end. With the counter-based generator keyed by
(seed, pass, pixel) the union of samples is the same for every G; only the order of the floating
point additions differs.

One process per GPU; `torch.distributed` provides the collective (backend "nccl" = RCCL over xGMI
on GPUs, "gloo" in the CPU tests).
"""


def partition_passes(first_pass, n_passes, rank, world):
    """Share of rank `rank` in passes [first_pass, first_pass + n_passes): (first, count, stride)
    as ubdpt_render_passes takes them."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank / world size")
    count = (n_passes - rank + world - 1) // world if n_passes > rank else 0
    return first_pass + rank, count, world


def reduce_sum(sum_tensor, dist=None, dst=0):
    """Adds every rank's sum buffer into rank `dst`'s (in place). `dist` = torch.distributed or None
    for a single process."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        if sum_tensor.is_cuda and dist.get_backend() == "gloo":  # CPU rehearsal of the GPU path: reduce through the host
            host = sum_tensor.cpu()
            dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM)
            sum_tensor.copy_(host)
        else:
            dist.reduce(sum_tensor, dst=dst, op=dist.ReduceOp.SUM)
    return sum_tensor

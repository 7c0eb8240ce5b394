"""Multi-GPU partition of the hot path: split the samples-per-pixel range across ranks.

Every (pixel, sample) path is independent and its random numbers are keyed by the GLOBAL sample
index, so rank g of G renders global samples [g*S/G, (g+1)*S/G) of ALL pixels into its own float
accumulation buffer (scene replicated per GPU), and ONE all-reduce(sum) of the buffers closes the
job (RCCL over xGMI when the backend is "nccl"; 1080p: 33.2 MB per rank). The set of samples is the
same for every G; images differ across G only by float summation order.
The reference has no multi-GPU path (reference: src/main.cpp:31-32 picks device 0).

ORDERING. ptx_render_samples is asynchronous on the context's own non-blocking HIP stream, which is
not ordered against torch's or RCCL's streams. The collective must not start before the kernels
that add into `accum` have finished: every function here calls `sync_fn` (ptx_render_sync for the
HIP renderer) between rendering and the collective. Passing sync_fn=None is only right for a
renderer that is synchronous (the CPU oracle in the gloo tests).
"""


def sample_range(rank, world, total_samples, first_sample=0):
    """Contiguous, disjoint, exhaustive: returns (first, count) of rank's share of
    [first_sample, first_sample + total_samples)."""
    if world < 1 or not (0 <= rank < world) or total_samples < 0:
        raise ValueError(f"bad partition rank={rank} world={world} total={total_samples}")
    lo = (rank * total_samples) // world
    hi = ((rank + 1) * total_samples) // world
    return first_sample + lo, hi - lo


def _rank_world(group=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def render_split(render_fn, accum, total_samples, first_sample=0, group=None, sync_fn=None):
    """render_fn(first, count) must ADD samples [first, first+count) into `accum` (a torch tensor,
    one per rank, zero on entry for a fresh image); sync_fn() must return only when those additions
    are complete (see ORDERING above). Collective: one all_reduce(SUM) on accum.
    Returns (first, count) this rank rendered."""
    import torch.distributed as dist
    rank, world = _rank_world(group)
    first, count = sample_range(rank, world, total_samples, first_sample)
    if count > 0:
        render_fn(first, count)
    if sync_fn is not None:
        sync_fn()
    if world > 1:
        dist.all_reduce(accum, op=dist.ReduceOp.SUM, group=group)
    return first, count


def render_progressive(render_fn, accum, total_samples, every, on_image=None, group=None, sync_fn=None):
    """Progressive variant (the 4K configuration): the job is cut into rounds of
    `every` GLOBAL samples; in each round rank g renders its share of that round's sample range into
    its own `accum` (which keeps holding only this rank's samples), and after each round a COPY of
    accum is all-reduced so that every rank - rank 0 in particular - holds the image of all samples
    rendered so far. on_image(image_tensor, samples_done) is called on every rank after each round
    (the tensor is reused: consume it before returning). The sample set is [0, total_samples) for
    every world size and every `every`. Returns the reduced image of the last round."""
    import torch.distributed as dist
    if every < 1:
        raise ValueError("every must be >= 1")
    rank, world = _rank_world(group)
    image = accum.clone()
    done = 0
    while done < total_samples:
        n = min(every, total_samples - done)
        first, count = sample_range(rank, world, n, done)
        if count > 0:
            render_fn(first, count)
        if sync_fn is not None:
            sync_fn()
        image.copy_(accum)
        if world > 1:
            dist.all_reduce(image, op=dist.ReduceOp.SUM, group=group)
        done += n
        if on_image is not None:
            on_image(image, done)
    return image

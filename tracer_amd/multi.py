"""Multi-GPU partition of a render: by sample index, one process per GPU, ONE sum-reduce of the
film buffers at the end. The reference has no multi-GPU path; per-sample
radiance is independent of the GPU count because the RNG is keyed by the absolute
(pixel, iteration), so the only coupling between ranks is this final sum.
"""


def rank_iterations(total_iters, rank, world):
    """Iterations {i in 1..total : (i-1) % world == rank} as (iter_first, iter_stride, count)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    if total_iters < 0:
        raise ValueError("bad iteration count")
    count = (total_iters - rank + world - 1) // world if total_iters > rank else 0
    return rank + 1, world, count


def tile_rows(height, rank, world, align=1):
    """Optional image-tile partition: contiguous row range [y0, y1) of rank (rows aligned)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    units = (height + align - 1) // align
    lo = units * rank // world * align
    hi = min(height, units * (rank + 1) // world * align)
    return lo, hi


def reduce_film(film, dist, dst=0):
    """The path's one collective: element-wise SUM of the per-rank films onto rank `dst`
    (RCCL when `film` is a GPU tensor and the backend is "nccl"; gloo on CPU tensors)."""
    if dist is not None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
    return film


def _torch_stream_sync(film):
    """Wait for everything torch has queued on its current stream of `film`'s device (no-op for CPU tensors)."""
    if getattr(film, "is_cuda", False):
        import torch
        torch.cuda.current_stream(film.device).synchronize()


def _shares_torch_stream(tracer, film):
    """True when the tracer launches on the very stream torch is using for `film`: the stream then orders
    zero_() -> render -> reduce -> tonemap by itself and no host-side waiting is needed."""
    if not getattr(film, "is_cuda", False):
        return False
    import torch
    handle = getattr(tracer, "stream_handle", None)
    return bool(handle) and handle == torch.cuda.current_stream(film.device).cuda_stream


def render_partitioned(tracer, cam, seed, total_iters, film, rank, world, dist, by_tile=False, height=None, timing=None):
    """Render this rank's share of `total_iters` iterations into `film` (a torch tensor bound to
    the tracer with Tracer.bind_film), sum-reduce to rank 0 and tonemap there. The share is a set of
    sample indices (default) or, with by_tile, a band of image rows rendered for every iteration; the
    films are zeroed first, so the same sum assembles disjoint tiles.

    Ordering. film.zero_() and the reduce run on torch's current stream, the render and the tonemap on the
    tracer's. Unless the two are the same stream (Tracer(stream=torch_stream.cuda_stream) with that stream
    current) the function orders them itself: it waits for the zero before rendering, for the render before
    the reduce, and for the reduce before the tonemap.

    With more than one rank the ranks' own display transforms are switched off for the render (only rank 0 needs an
    LDR image, and only of the reduced film: it runs the transform after the reduce).

    timing: an optional list; for a GPU film a pair of torch events recorded around the reduce on torch's current stream
    is appended to it (the caller reads elapsed_time after synchronising: on the root this is the wait for the slowest
    rank plus the transfer)."""
    same = _shares_torch_stream(tracer, film)
    film.zero_()
    if not same:
        _torch_stream_sync(film)
    multi = world > 1 and hasattr(tracer, "set_skip_tonemap")
    if multi:
        tracer.set_skip_tonemap(True)
    try:
        if by_tile:
            y0, y1 = tile_rows(tracer.h if height is None else height, rank, world)
            if y1 > y0:
                tracer.set_tile(y0, y1 - y0)
                try:
                    tracer.pathtrace(cam, seed, 1, total_iters)
                finally:
                    tracer.set_tile(0, 0)
        else:
            first, stride, count = rank_iterations(total_iters, rank, world)
            if count:
                tracer.pathtrace(cam, seed, first, count, iter_stride=stride)
    finally:
        if multi:
            tracer.set_skip_tonemap(False)
    if not same:
        tracer.synchronize()
    events = None
    if timing is not None and getattr(film, "is_cuda", False) and world > 1:
        import torch
        events = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        events[0].record(torch.cuda.current_stream(film.device))
    reduce_film(film, dist, 0)
    if events is not None:
        import torch
        events[1].record(torch.cuda.current_stream(film.device))
        timing.append(events)
    if rank == 0 and world > 1:          # (a single rank's pathtrace call has already run the display transform)
        if not same:
            _torch_stream_sync(film)
        tracer.tonemap(total_iters)
    return film

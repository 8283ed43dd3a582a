"""Row-stripe decomposition of a frame across GPUs (one process per GPU).

The reference parallelises a render by contiguous row bands on threads
(call_invocation_parallel, mathmap_common.c:991-1003); the same band formula is used
here across ranks.  Pixels are independent, so the data path needs no collective:
every rank holds a replica of the (small) input image and renders rows
[H*g/G, H*(g+1)/G) of the output into its own HBM.  Only gathering the finished stripes
on one rank (for writing a file) communicates.

The Gaussian blur's vertical pass couples rows, but because the input is replicated the halo
a stripe needs (ceil(22.7 sigma) rows above and below, after which the recurrences' start-up
error is below half an ulp of their f64 sums) is re-rendered locally instead of exchanged:
`native_row_margin` lets a stripe render fill only its own window of the blur map
(native_filters.hip gaussian_blur; tests/test_gpu_parity.py::
test_gauss_row_stripes_with_local_halo_equal_full_frame checks bit-equality with the full frame).
"""


def animation_frame_t(frame, num_frames):
    """t of frame `frame` of a `num_frames`-frame animation, as the reference's command line computes
    it: (float)frame / (float)num_frames in float arithmetic (mathmap_cmdline.c:835)."""
    import numpy as np
    return float(np.float32(frame) / np.float32(num_frames))


def stripe_rows(height, rank, world):
    """Rows [lo, hi) of rank `rank` out of `world`: the reference's band formula."""
    lo = height * rank // world
    hi = height * (rank + 1) // world
    return lo, hi


def render_stripe(inv, out_ptr, rank, world, t=0.0, frame=0, stream=0, bpp=4, native_row_margin=None):
    """Renders this rank's stripe of the frame into device memory at out_ptr (first row of
    the stripe).  Coordinates are computed from absolute rows, so the union of all ranks'
    stripes is bit-identical to a single-GPU render.  `native_row_margin` (rows, or None):
    for filters that sample a native-filter map no further than that many rows from the output
    row -- 0 for `blurred(xy)` -- the map is computed for the stripe's window only."""
    lo, hi = stripe_rows(inv.height, rank, world)
    if native_row_margin is not None:
        inv.set_native_row_margin(native_row_margin)
    inv.render_rows(out_ptr, lo, hi, t=t, frame=frame, bpp=bpp, stream=stream)
    return lo, hi


def gather_stripes(local_stripe, height, rank, world, group=None):
    """Collects the stripes on rank 0 as one [H, W, C] tensor (torch.distributed gather; RCCL
    over xGMI on GPUs, gloo on CPU tensors).  Stripe heights may differ by one row."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local_stripe
    w, c = local_stripe.shape[1], local_stripe.shape[2]
    max_rows = max(stripe_rows(height, r, world)[1] - stripe_rows(height, r, world)[0] for r in range(world))
    padded = torch.zeros((max_rows, w, c), dtype=local_stripe.dtype, device=local_stripe.device)
    padded[: local_stripe.shape[0]] = local_stripe
    bufs = [torch.empty_like(padded) for _ in range(world)] if rank == 0 else None
    dist.gather(padded, bufs, dst=0, group=group)
    if rank != 0:
        return None
    parts = []
    for r in range(world):
        lo, hi = stripe_rows(height, r, world)
        parts.append(bufs[r][: hi - lo])
    return torch.cat(parts, dim=0)


def replicate_input(image, height, width, rank, world, src=0, group=None):
    """Replicates the packed RGBA8 input image (a [H, W] int32/uint32 tensor on rank `src`) on every
    rank as scatter + all-gather: the source sends each rank a different 1/world slice (one slice
    per xGMI link instead of the whole image down a ring), then the slices are all-gathered.
    xGMI is point to point, 7 links x ~153 GB/s per GPU: a 268 MB frame takes ~0.3 ms this way
    against ~1.8 ms for a single-link broadcast.  Returns the full image on every rank
    (RCCL on GPU tensors, gloo on CPU tensors)."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return image
    n = height * width
    chunk = (n + world - 1) // world
    device = image.device if rank == src else None
    dtype = image.dtype if rank == src else None
    meta = [device, dtype]
    dist.broadcast_object_list(meta, src=src, group=group)
    device, dtype = (meta[0] if rank != src else image.device), meta[1]
    if rank != src and device.type == "cuda":
        device = torch.device("cuda", torch.cuda.current_device())
    mine = torch.empty(chunk, dtype=dtype, device=device)
    if rank == src:
        flat = torch.zeros(chunk * world, dtype=dtype, device=device)
        flat[:n] = image.reshape(-1)
        dist.scatter(mine, list(flat.view(world, chunk).unbind(0)), src=src, group=group)
    else:
        dist.scatter(mine, None, src=src, group=group)
    parts = [torch.empty(chunk, dtype=dtype, device=device) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    return torch.cat(parts)[:n].view(height, width)

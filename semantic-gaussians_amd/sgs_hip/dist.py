"""Multi-GPU use of the rasteriser: one process per GPU, `torch.distributed` (backend "nccl" =
RCCL over xGMI on ROCm; "gloo" on CPU for tests).

The hot path shards without any data-path collective: frames are independent
given a read-only scene, so the scene is replicated and the VIEW list is split round-robin.
For scenes whose feature table does not fit one GPU the exact alternative is CHANNEL sharding:
geometry is replicated, rank r holds feature columns [r*C/w, (r+1)*C/w), every rank runs the
identical preprocess/sort and blends its own channel slice -- bit-identical to the single-GPU
render with zero reduction; one all_gather only if a single rank needs the full map.

GAUSSIAN sharding (scenes whose geometry does not fit either) is the one
variant with a real exchange step: every rank renders its shard into a per-pixel partial
(A in R^C, T), and partials combine with the associative, NON-commutative "over" operator
(A1, T1) o (A2, T2) = (A1 + T1*A2, T1*T2) in front-to-back shard order -- exact only for
depth-separable shards (view-space depth slabs).  The exchange is image-partitioned: rank g owns
a band of rows, receives the other ranks' partials for that band (grouped point-to-point
send/recv = an all-to-all over all xGMI links at once, not a ring), composites them in shard
order and adds bg*T; an optional all_gather rebuilds the full map.  A ring all-reduce would be
the wrong operator (it sums) and single-link bound.
"""
import time

import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _host_transport_fence(t):
    """gloo moves bytes with the CPU and knows nothing about HIP streams: a device tensor handed to it must be COMPLETE in
    memory first (round 4: the one-device gloo test of the Gaussian-sharded render exchanged bands the render kernels were
    still writing -- the old O(1) slack of that test hid it).  Under RCCL the process group orders its kernels behind the
    current stream itself; nothing to do."""
    if t.is_cuda and dist.is_initialized() and dist.get_backend() == "gloo":
        torch.cuda.synchronize(t.device)


def shard_views(n_views, rank=None, world_size=None):
    """Round-robin view indices of this rank."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    return list(range(rank, n_views, world_size))


def render_views_sharded(render_fn, views, gather_to=None):
    """Each rank renders views[rank::world].  No collective touches the render itself.

    gather_to=None (default since round 3 -- it used to be 0): results stay on the GPU that produced them -- returns
    {view index: tensor} of this rank's views (a 3.3 GB feature map per view at BASELINE config 4 has no business crossing
    PCIe or being pickled).  Callers that relied on the old default pass gather_to=0 explicitly.
    gather_to=r: rank r additionally receives the other ranks' results as TENSORS (grouped point-to-point receives
    straight into device memory under RCCL; every view must have the same shape and dtype) and returns the list in
    view order; the other ranks return None.  An empty view list returns {} / [] / None without any collective."""
    rank, w = world()
    if len(views) == 0:
        return {} if gather_to is None else ([] if rank == gather_to or w == 1 else None)
    mine = {i: render_fn(views[i]) for i in shard_views(len(views), rank, w)}
    if gather_to is None:
        return mine
    if w == 1:
        return [mine[i] for i in range(len(views))]
    # shape / dtype of a view: known to every rank that rendered one (rank 0 always has: views is not empty); agree on it
    # with one tiny object collective
    meta = [None] * w
    first = next(iter(mine.values())) if mine else None
    dist.all_gather_object(meta, None if first is None else (tuple(first.shape), str(first.dtype).replace("torch.", ""), first.device.type))
    shape, dtype, devtype = next(m for m in meta if m is not None)
    ops, recv = [], {}
    if rank == gather_to:
        device = first.device if first is not None else (torch.device("cuda", torch.cuda.current_device()) if devtype == "cuda" else torch.device("cpu"))
        for i in range(len(views)):
            src = i % w
            if src != rank:
                recv[i] = torch.empty(shape, dtype=getattr(torch, dtype), device=device)
                ops.append(dist.P2POp(dist.irecv, recv[i], src))
    else:
        for i, t in sorted(mine.items()):
            ops.append(dist.P2POp(dist.isend, t.contiguous(), gather_to))
    if ops:
        if first is not None:
            _host_transport_fence(first)
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if rank != gather_to:
        return None
    return [mine[i] if i % w == rank else recv[i] for i in range(len(views))]


_SIDE_STREAMS = {}


def render_views_pipelined(render_fn, views, in_flight=2, device=None):
    """Render `views` on one GPU keeping `in_flight` of them in flight, one per HIP stream, from the
    calling thread: while one view waits for its num_rendered read-back and runs its small binning
    kernels, another view's blend keeps the memory system busy (cfg3 on MI355X: 2.1 -> 1.8 ms per view).

    `render_fn(view, slot)` must enqueue on the CURRENT stream (the rasteriser does) and use
    per-slot resources for anything it keeps across calls (e.g. `ScratchPool` number `slot`).  It may return
    `raster.rasterize_forward_deferred(...)` handles: then the host never waits for the GPU while it enqueues
    (cfg3: 1.47 -> see DESIGN.md 7 ms per view with four in flight); they are resolved here, one frame per slot behind.
    Results are returned in view order after a device synchronize."""
    if not torch.cuda.is_available():
        return [render_fn(v, 0) for v in views]
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    n = max(1, min(in_flight, len(views)))
    # the side streams are created once per device and reused: scratch pools are keyed by stream, so fresh
    # streams on every call would leave a fresh set of state buffers behind each time
    side = _SIDE_STREAMS.setdefault(device, [])
    while len(side) < n - 1:
        side.append(torch.cuda.Stream(device))
    streams = [torch.cuda.current_stream(device)] + side[:n - 1]
    for st in streams[1:]:
        st.wait_stream(streams[0])   # inputs produced on the caller's stream are visible to the others
    out = []
    pending = [None] * n   # per slot: index of a deferred-count forward (raster.rasterize_forward_deferred) not yet resolved
    for i, v in enumerate(views):
        slot = i % n
        if pending[slot] is not None:   # before the slot's state buffers are handed to the next frame
            out[pending[slot]] = out[pending[slot]].result()
            pending[slot] = None
        with torch.cuda.stream(streams[slot]):
            r = render_fn(v, slot)
        out.append(r)
        if hasattr(r, "result") and hasattr(r, "layout_count"):
            pending[slot] = i
    for j in pending:
        if j is not None:
            out[j] = out[j].result()
    for st in streams[1:]:
        streams[0].wait_stream(st)
    torch.cuda.synchronize(device)
    return out


def channel_slice(C, rank=None, world_size=None):
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    if C % world_size:
        raise ValueError(f"C={C} must be divisible by the world size {world_size}")
    per = C // world_size
    return rank * per, (rank + 1) * per


def render_channel_sharded(render_fn, features, bg, all_gather=True):
    """Exact channel sharding.  `render_fn(features_slice, bg_slice) -> (c,H,W)` renders this
    rank's columns; with all_gather=True every rank returns the full (C,H,W) map."""
    rank, w = world()
    lo, hi = channel_slice(features.shape[1], rank, w)
    part = render_fn(features[:, lo:hi].contiguous(), bg[lo:hi].contiguous())
    if w == 1 or not all_gather:
        return part
    parts = [torch.empty_like(part) for _ in range(w)]
    dist.all_gather(parts, part.contiguous())
    return torch.cat(parts, dim=0)


def band_rows(H, rank=None, world_size=None):
    """Rows [lo, hi) of the image band owned by `rank` (16-row tile granularity where possible)."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    tiles = (H + 15) // 16
    lo = min(H, 16 * (tiles * rank // world_size))
    hi = min(H, 16 * (tiles * (rank + 1) // world_size))
    return lo, hi


def composite_over(partials, bg=None):
    """Front-to-back "over" of [(A (C,h,W), T (h,W)), ...]; adds bg*T_total when bg is given.  -> (out, T_total).
    Device tensors: ONE HIP kernel that reads every partial once (C-ABI sgs_composite_over); host tensors (the gloo
    tests, whose renderer is injected): the same chain in torch."""
    a0 = partials[0][0]
    if a0.is_cuda and len(partials) <= 16:
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        S = len(partials)
        Cn, h, W = a0.shape
        keep = [(a.contiguous(), t.contiguous()) for a, t in partials]
        if any(x.dtype != torch.float32 for pair in keep for x in pair):
            raise RuntimeError("partials must be float32")
        out = torch.empty((Cn, h, W), dtype=torch.float32, device=a0.device)
        t_out = torch.empty((h, W), dtype=torch.float32, device=a0.device)
        if (h * W) % 4 == 0 and all((x.data_ptr() & 15) == 0 for pair in keep for x in pair):
            pa = (C.c_void_p * S)(*[a.data_ptr() for a, _ in keep])
            pt = (C.c_void_p * S)(*[t.data_ptr() for _, t in keep])
            bgc = None if bg is None else bg.to(device=a0.device, dtype=torch.float32).contiguous()
            with torch.cuda.device(a0.device):
                rc = lib.sgs_composite_over(S, pa, pt, None if bgc is None else C.c_void_p(bgc.data_ptr()),
                                            C.c_void_p(out.data_ptr()), C.c_void_p(t_out.data_ptr()), Cn, h, W,
                                            C.c_void_p(torch.cuda.current_stream(a0.device).cuda_stream))
            _lib.check(rc, "composite_over failed")
            return out, t_out
    out = partials[0][0].clone()
    t_acc = partials[0][1].clone()
    for a, t in partials[1:]:
        out += t_acc.unsqueeze(0) * a
        t_acc = t_acc * t
    if bg is not None:
        out += bg.reshape(-1, 1, 1) * t_acc.unsqueeze(0)
    return out, t_acc


def render_gaussian_sharded(render_partial_fn, bg, order=None, all_gather=True):
    """Gaussian-sharded render of one view.  Rank r holds shard r of the Gaussians;
    `render_partial_fn() -> (A (C,H,W) rendered with a ZERO background, T (H,W) final transmittance)`
    renders the local shard.  `order` lists the ranks front to back for this view (default
    0..w-1): the shards must be depth-separable in that order for the result to equal the
    single-GPU render (up to fp32 rounding and the per-shard instead of global T < 1e-4 stop).
    Returns the full (C,H,W) map (all_gather=True) or this rank's band of rows."""
    rank, w = world()
    a, t = render_partial_fn()
    # A as a LIST of the w image bands (raster.render_partial(..., bands=w): the kernels wrote the map band-major): every band is already
    # one contiguous message.  A (C,H,W) tensor is sliced and each band copied once (the transport needs contiguous buffers).
    banded = isinstance(a, (list, tuple))
    if w == 1:
        return composite_over([(torch.cat(list(a), dim=1) if banded else a, t)], bg)[0]
    if banded and len(a) != w:
        raise ValueError(f"the partial comes in {len(a)} bands, the job has {w} ranks")
    order = list(range(w)) if order is None else list(order)
    H = t.shape[0]
    a0 = a[0] if banded else a
    Cn, Wd = a0.shape[0], a0.shape[2]
    # image-partitioned all-to-all of the (A, T) partials: grouped point-to-point send / recv
    mine = band_rows(H, rank, w)
    recv = {}
    ops, keep = [], []
    for peer in range(w):
        lo, hi = band_rows(H, peer, w)
        sa = a[peer] if banded else a[:, lo:hi].contiguous()
        st = t[lo:hi]   # (rows of an (H,W) plane: contiguous as it is)
        if banded and tuple(sa.shape) != (Cn, hi - lo, Wd):
            raise ValueError(f"band {peer} has shape {tuple(sa.shape)}, expected {(Cn, hi - lo, Wd)}")
        if peer == rank:
            recv[rank] = (sa, st)
            continue
        ra = torch.empty((Cn, mine[1] - mine[0], Wd), dtype=a0.dtype, device=a0.device)
        rt = torch.empty((mine[1] - mine[0], Wd), dtype=t.dtype, device=t.device)
        keep += [sa, st]
        recv[peer] = (ra, rt)
        if sa.numel():
            ops += [dist.P2POp(dist.isend, sa, peer), dist.P2POp(dist.isend, st, peer)]
        if ra.numel():
            ops += [dist.P2POp(dist.irecv, ra, peer), dist.P2POp(dist.irecv, rt, peer)]
    if ops:
        _host_transport_fence(a0)   # (the contiguous send copies above are complete, too)
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    band, _ = composite_over([recv[r] for r in order], bg)
    if not all_gather:
        return band
    # every rank needs every band: an all-gather of bands of (possibly) different heights, done as grouped point-to-point
    # operations on the exact sizes -- no zero padding, no staging copy of the band, every xGMI link carries one band each
    # way (uniform bands could use all_gather_into_tensor; 61 tile rows over 8 ranks are not uniform)
    sizes = [band_rows(H, r, w) for r in range(w)]
    band = band.contiguous()
    _host_transport_fence(band)
    parts = [band if r == rank else torch.empty((band.shape[0], sizes[r][1] - sizes[r][0], band.shape[2]), dtype=band.dtype, device=band.device)
             for r in range(w)]
    ops = []
    for peer in range(w):
        if peer == rank:
            continue
        if band.numel():
            ops.append(dist.P2POp(dist.isend, band, peer))
        if parts[peer].numel():
            ops.append(dist.P2POp(dist.irecv, parts[peer], peer))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return torch.cat(parts, dim=1)


def timed_steps(step_fn, steps, warmup, sync_fn=None):
    """The bench contract: `warmup` untimed steps, then exactly `steps` steps bracketed by a
    barrier (+ device sync) on both sides; returns the MAX over ranks of the elapsed seconds."""
    rank, w = world()

    def barrier():
        if w > 1:
            dist.barrier()
        if sync_fn is not None:
            sync_fn()

    for _ in range(warmup):
        step_fn()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    barrier()
    t = time.perf_counter() - t0
    if w > 1:
        tt = torch.tensor([t], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            tt = tt.cuda()
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t = float(tt.item())
    return t

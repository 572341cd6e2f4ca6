"""Torch-tensor front end of the C-ABI: the counterpart of the reference's C++ glue
(CR/rasterize_points.cu:28-223, RR/rasterize_points.cu, SK/spatial.cu:15-26).

PyTorch is used for device memory and streams only: outputs and the three opaque state
buffers are torch tensors (caching allocator, current stream), everything else happens
inside libsgs_hip.so.
"""
import ctypes as C
import os
import threading
import warnings

import torch

from . import _lib


def _ptr(t, name, device, dtype=torch.float32):
    """Device pointer of an input tensor; empty tensor -> NULL ("not provided",
    CR/channel_rasterization/__init__.py:266-276)."""
    if t is None or t.numel() == 0:
        return None, None
    if t.dtype != dtype:
        raise RuntimeError(f"expected scalar type {dtype} for {name} but found {t.dtype}")
    if t.device != device:
        raise RuntimeError(f"{name} is on {t.device}, expected {device}")
    t = t.contiguous()   # non-contiguous inputs are legal (rasterize_points.cu:99-117)
    return t.data_ptr(), t


class ScratchPool:
    """Grow-only per-device scratch for the three opaque state buffers.

    Inference callers (torch.no_grad, nothing to backpropagate) do not need the buffers to
    survive the call, so re-allocating ~2 GB of scratch per frame through the caching
    allocator is pure overhead (and makes it thrash: blocks of 2.6 GB / 1.6 GB / 0.1 GB are
    split and re-split for many frames before it settles).  A pool keeps them resident.
    Buffers handed out from a pool are only valid until the next forward on that pool AND stream:
    entries are keyed by (device, current stream, buffer), so forwards running concurrently on
    different streams (sgs_hip.dist.render_views_pipelined) never share scratch."""

    def __init__(self):
        self._t = {}

    def get(self, device, key, nbytes):
        stream = torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0
        k = (str(device), stream, key)
        t = self._t.get(k)
        if t is None or t.numel() < nbytes:
            nb = int(nbytes * 1.25) if t is not None else int(nbytes)   # amortise growth
            t = torch.empty(nb, dtype=torch.uint8, device=device)
            self._t[k] = t
        return t

    def clear(self):
        self._t.clear()


INFERENCE_POOL = ScratchPool()


class _Buffers:
    """Growable byte buffers handed to the library through the sgs_alloc_fn callback
    (the reference's resizeFunctional, CR/rasterize_points.cu:28-36).

    The C callback itself is ONE process-wide ctypes trampoline; which _Buffers / which buffer it serves travels
    in the callback's `user` word.  (A fresh CFUNCTYPE closure per buffer per frame makes libffi map and unmap
    executable pages all the time: a ~10 ms host stall every few dozen frames on a 256-thread host.)"""

    _live = {}
    _next = 1
    _lock = threading.Lock()   # one host thread per GPU is a legal way to drive several devices: handles must be unique
    KEYS = ("g", "b", "i", "s")

    def __init__(self, device, pool=None):
        self.device = device
        self.pool = pool
        self.tensors = {}
        with _Buffers._lock:
            self.handle = _Buffers._next
            _Buffers._next += 1
            _Buffers._live[self.handle] = self

    def release(self):
        _Buffers._live.pop(self.handle, None)

    def _alloc(self, key, nbytes):
        try:
            if self.pool is not None:
                t = self.pool.get(self.device, key, int(nbytes))
            else:
                t = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
            self.tensors[key] = t
            return t.data_ptr()
        except Exception:   # noqa: BLE001 - reported as SGS_EALLOC by the library
            return None

    def callback(self, key):
        """(function pointer, user word) pair for buffer `key`."""
        return _TRAMPOLINE, C.c_void_p(self.handle * 4 + _Buffers.KEYS.index(key))

    def get(self, key):
        t = self.tensors.get(key)
        if t is None:
            t = torch.empty(0, dtype=torch.uint8, device=self.device)
        return t


def _trampoline(user, nbytes):
    user = int(user or 0)
    bufs = _Buffers._live.get(user >> 2)
    if bufs is None:
        return None
    return bufs._alloc(_Buffers.KEYS[user & 3], nbytes)


_TRAMPOLINE = _lib.ALLOC_FN(_trampoline)


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


# Opt-in: allocate the output planes with rows padded to a multiple of this many pixels and return the (C,H,W) VIEW of
# them (non-contiguous when W is not a multiple).  32 makes every tile pair whole 128-byte lines for any width
# (cfg4's 1297: the accumulate kernel's stores run at the full-line rate instead of the partial-line rate).
OUTPUT_PITCH_ALIGN = int(os.environ.get("SGS_OUTPUT_PITCH_ALIGN", "0") or 0)
STRICT_BG = os.environ.get("SGS_STRICT_BG", "0") not in ("", "0")
_bg_warned = False


def _check_bg(bg, Cn):
    """A background shorter than num_channels: the reference reads past its end
    (CR/cuda_rasterizer/forward.cu:373; view_viser.py:82-83,303-313 passes a 3-vector with C ~ 20).
    Here the missing channels are defined as 0 -- the script keeps running and nothing is read out of
    bounds -- with one warning per process; SGS_STRICT_BG=1 makes it an error instead."""
    global _bg_warned
    if bg is None or bg.numel() >= Cn:
        return bg
    if STRICT_BG:
        raise RuntimeError(
            f"bg has {bg.numel()} entries but num_channels={Cn} (the reference reads out of "
            "bounds here, CR/cuda_rasterizer/forward.cu:373)")
    if not _bg_warned:
        warnings.warn(f"bg has {bg.numel()} entries but num_channels={Cn}: missing channels are taken as 0 "
                      "(the reference reads out of bounds here; set SGS_STRICT_BG=1 to make this an error)")
        _bg_warned = True
    pad = torch.zeros(Cn, dtype=bg.dtype, device=bg.device)
    pad[: bg.numel()] = bg.reshape(-1)
    return pad


class DeferredForward:
    """A forward whose instance counts stayed on the device (SGS_OPT_DEFER_COUNT, include/sgs_raster.h): the call
    that made it did not wait for the GPU.  result() -- call it on the host thread that made the forward, before the
    outputs are used and before the next forward on the same stream -- waits for the counts (they arrive after the frame's scan, long before its blend) and returns
    the same tuple rasterize_forward returns, with the true num_rendered; a frame that did not fit this stream's
    capacity guess is rendered again the ordinary way (the guess has grown by then).  Inference only: the
    binning buffer is laid out for `layout_count` (>= num_rendered) entries, which rasterize_backward cannot know."""

    def __init__(self, args, kwargs, stream, out):
        self._args, self._kwargs, self._stream, self._out = args, kwargs, stream, out
        self.layout_count = out[0]
        self.retried = False

    def result(self):
        if self._args is None:
            return self._out
        lib = _lib.load()
        n = C.c_int(0)
        # the library keys its per-stream context on (current device, stream): resolve under the forward's device
        with torch.cuda.device(self._stream.device):
            rc = lib.sgs_forward_result(C.c_void_p(self._stream.cuda_stream), 1, C.byref(n))
        if rc == _lib.ERETRY:
            with torch.cuda.device(self._stream.device), torch.cuda.stream(self._stream):
                self._out = rasterize_forward(*self._args, **self._kwargs)
            self.retried = True
            self.layout_count = self._out[0]
        else:
            _lib.check(rc, "rasterize_gaussians failed")
            self._out = (int(n.value),) + tuple(self._out[1:])
        self._args = self._kwargs = None
        return self._out


def rasterize_forward_deferred(*args, **kwargs):
    """rasterize_forward without the host waiting for the GPU: -> DeferredForward.  The first frame of a stream is an
    ordinary (blocking) forward -- it is where the capacity guesses come from."""
    lib = _lib.load()
    if args[1].ndimension() != 2 or args[1].size(1) != 3:   # (rasterize_forward's own checks, in its order)
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    if not args[1].is_cuda:
        raise RuntimeError("means3D must be a GPU tensor (there is no CPU path)")
    dev = args[1].device
    stream = torch.cuda.current_stream(dev)
    sp = C.c_void_p(stream.cuda_stream)
    if args[1].size(0) == 0:   # nothing to render: the library is not called at all, there is no count to wait for
        kwargs.pop("_defer_mode", None)
        return DeferredForward(None, None, stream, rasterize_forward(*args, **kwargs))
    with torch.cuda.device(dev):
        prev = lib.sgs_stream_set_option(sp, _lib.OPT_DEFER_COUNT, int(kwargs.pop("_defer_mode", 1)))
        try:
            out = rasterize_forward(*args, **kwargs)
        finally:
            lib.sgs_stream_set_option(sp, _lib.OPT_DEFER_COUNT, -1 if prev == 0x7fffffff else prev)
    return DeferredForward(args, kwargs, stream, out)


def rasterize_forward_inference(*args, **kwargs):
    """rasterize_forward for a frame nobody will differentiate (what GaussianRasterizer does under torch.no_grad()): same
    arguments, same tuple, num_rendered known when it returns -- but the whole frame is enqueued against the stream's capacity
    guess BEFORE the host waits for the counts, so the GPU's timeline has no read-back hole (a frame that outgrew
    the guess is rendered again).  The state buffers are laid out for the capacity: not for rasterize_backward."""
    return rasterize_forward_deferred(*args, **kwargs).result()


def rasterize_forward(background, means3D, colors, opacity, scales, rotations, scale_modifier,
                      cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height,
                      image_width, sh, degree, campos, prefiltered, debug, num_channels,
                      want_depth, pool=None, norm_plane=False, out_bands=0):
    """RasterizeGaussiansCUDA (CR/rasterize_points.cu:38-121; RR variant returns depth too).
    Returns (num_rendered, out_color, radii, geomBuffer, binningBuffer, imgBuffer, out_depth).
    pool: optional ScratchPool for the three state buffers (inference; see ScratchPool).
    norm_plane: out_color is the (H,W) plane sum_c render[c]^2 instead of the (C,H,W) render (SGS_OPT_NORM_PLANE,
    include/sgs_raster.h; C % 128 == 0, inference only).
    out_bands = n > 1: out_color is a LIST of n contiguous (C, rows_b, W) tensors, the image bands of sgs_hip.dist.band_rows (views of one
    buffer the kernels wrote band-major, SGS_OPT_OUT_BANDS; C % 128 == 0, no depth): each band is one message of an image-partitioned exchange."""
    lib = _lib.load()
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    if not means3D.is_cuda:
        raise RuntimeError("means3D must be a GPU tensor (there is no CPU path)")
    dev = means3D.device
    P, H, W, Cn = means3D.size(0), int(image_height), int(image_width), int(num_channels)
    keep = []
    with torch.cuda.device(dev):
        bufs = _Buffers(dev, pool)
        radii = torch.empty(P, dtype=torch.int32, device=dev)
        depth = None
        def band_views(flat):
            from .dist import band_rows   # the one definition of the bands (the kernels' sgs_band_of restates it: tests/test_multigpu.py)
            views = []
            for b in range(out_bands):
                lo, hi = band_rows(H, b, out_bands)
                views.append(flat[Cn * W * lo:Cn * W * hi].view(Cn, hi - lo, W))
            return views
        if out_bands > 1 and (Cn % 128 or want_depth or norm_plane):
            raise RuntimeError("out_bands needs a multiple of 128 channels, no depth plane and no norm plane")
        if P == 0:
            # reference returns zero-filled outputs without touching bg (rasterize_points.cu:85)
            color = torch.zeros((H, W) if norm_plane else (Cn, H, W), dtype=torch.float32, device=dev)
            if out_bands > 1:
                color = band_views(color.view(-1))
            if want_depth:
                depth = torch.zeros(1, H, W, dtype=torch.float32, device=dev)
            bufs.release()
            return 0, color, radii, bufs.get("g"), bufs.get("b"), bufs.get("i"), depth
        pitch = W
        if norm_plane:
            if Cn % 128 or want_depth:
                raise RuntimeError("norm_plane needs a multiple of 128 channels and no depth plane")
        elif OUTPUT_PITCH_ALIGN > 1 and W % OUTPUT_PITCH_ALIGN and Cn >= 128 and not want_depth and out_bands <= 1:
            pitch = -(-W // OUTPUT_PITCH_ALIGN) * OUTPUT_PITCH_ALIGN
        color = torch.empty((H, W) if norm_plane else (Cn, H, pitch), dtype=torch.float32, device=dev)   # fully overwritten
        if want_depth:
            depth = torch.empty(1, H, W, dtype=torch.float32, device=dev)

        def p(t, name):
            ptr, kept = _ptr(t, name, dev)
            keep.append(kept)
            return ptr

        bg = _check_bg(background, Cn)
        M = sh.size(1) if (sh is not None and sh.numel() != 0) else 0
        # every argument is marshalled BEFORE the pitch is set (nothing between the set and the call can raise); the
        # library consumes the override in that call, the finally covers a failure inside ctypes itself
        args = (*bufs.callback("g"), *bufs.callback("b"), *bufs.callback("i"),
                P, int(degree), int(M), p(bg, "bg"), W, H, p(means3D, "means3D"), p(sh, "sh"),
                p(colors, "colors_precomp"), p(opacity, "opacities"), p(scales, "scales"),
                float(scale_modifier), p(rotations, "rotations"), p(cov3D_precomp, "cov3D_precomp"),
                p(viewmatrix, "viewmatrix"), p(projmatrix, "projmatrix"), p(campos, "campos"),
                float(tan_fovx), float(tan_fovy), int(bool(prefiltered)), Cn, color.data_ptr(),
                depth.data_ptr() if depth is not None else None, radii.data_ptr(), int(bool(debug)),
                _stream_ptr(dev))
        try:
            if pitch != W:
                lib.sgs_stream_set_option(_stream_ptr(dev), _lib.OPT_OUT_PITCH, pitch)
            if norm_plane:
                lib.sgs_stream_set_option(_stream_ptr(dev), _lib.OPT_NORM_PLANE, 1)
            if out_bands > 1:
                lib.sgs_stream_set_option(_stream_ptr(dev), _lib.OPT_OUT_BANDS, int(out_bands))
            rc = lib.sgs_rasterize_forward(*args)
        finally:
            bufs.release()
            if out_bands > 1:
                lib.sgs_stream_set_option(_stream_ptr(dev), _lib.OPT_OUT_BANDS, -1)
            if pitch != W:
                lib.sgs_stream_set_option(_stream_ptr(dev), _lib.OPT_OUT_PITCH, -1)
            if norm_plane:
                lib.sgs_stream_set_option(_stream_ptr(dev), _lib.OPT_NORM_PLANE, -1)
        if pitch != W:
            color = color[:, :, :W]
        num_rendered = _lib.check(rc, "rasterize_gaussians failed")
        if out_bands > 1:
            color = band_views(color.view(-1))
    return num_rendered, color, radii, bufs.get("g"), bufs.get("b"), bufs.get("i"), depth


def rasterize_backward(background, means3D, radii, colors, scales, rotations, scale_modifier,
                       cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                       sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, debug):
    """RasterizeGaussiansBackwardCUDA (CR/rasterize_points.cu:123-202) with the colour channel
    count taken from dL_dout_color (the reference hard-codes NUM_CHANNELS = 3)."""
    lib = _lib.load()
    dev = means3D.device
    P = means3D.size(0)
    Cn, H, W = dL_dout_color.size(0), dL_dout_color.size(1), dL_dout_color.size(2)
    M = sh.size(1) if (sh is not None and sh.numel() != 0) else 0
    opts = dict(dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        # the per-Gaussian gradients: ONE zero-filled allocation carved into the eight tensors (one fill kernel instead
        # of eight), each a contiguous view
        widths = (3, 3, 4, 1, 6, 3 * M, 3, 4)
        flat = torch.zeros(P * sum(widths), **opts)
        parts, off = [], 0
        for w in widths:
            parts.append(flat[off:off + P * w])
            off += P * w
        dL_dmeans3D, dL_dmeans2D = parts[0].view(P, 3), parts[1].view(P, 3)
        dL_dconic, dL_dopacity, dL_dcov3D = parts[2].view(P, 2, 2), parts[3].view(P, 1), parts[4].view(P, 6)
        dL_dsh, dL_dscales, dL_drotations = parts[5].view(P, M, 3), parts[6].view(P, 3), parts[7].view(P, 4)
        # the (P, C) colour gradient: the library clears it inside its first kernel (SGS_OPT_BWD_CLEARS_DCOLOR)
        dL_dcolors = torch.empty(P, Cn, **opts) if P != 0 else torch.zeros(P, Cn, **opts)
        if P != 0:
            sp = _stream_ptr(dev)
            keep = []

            def p(t, name, dtype=torch.float32):
                ptr, kept = _ptr(t, name, dev, dtype)
                keep.append(kept)
                return ptr

            bargs = (
                P, int(degree), int(M), int(R), p(background, "bg"), W, H, p(means3D, "means3D"),
                p(sh, "sh"), p(colors, "colors_precomp"), p(scales, "scales"),
                float(scale_modifier), p(rotations, "rotations"),
                p(cov3D_precomp, "cov3D_precomp"), p(viewmatrix, "viewmatrix"),
                p(projmatrix, "projmatrix"), p(campos, "campos"), float(tan_fovx),
                float(tan_fovy), p(radii, "radii", torch.int32),
                p(geomBuffer, "geomBuffer", torch.uint8),
                p(binningBuffer, "binningBuffer", torch.uint8),
                p(imageBuffer, "imageBuffer", torch.uint8), p(dL_dout_color, "dL_dout_color"),
                Cn, dL_dmeans2D.data_ptr(), dL_dconic.data_ptr(), dL_dopacity.data_ptr(),
                dL_dcolors.data_ptr(), dL_dmeans3D.data_ptr(), dL_dcov3D.data_ptr(),
                dL_dsh.data_ptr() if M else None, dL_dscales.data_ptr(),
                dL_drotations.data_ptr(), int(bool(debug)), sp)
            prev_clear = lib.sgs_stream_set_option(sp, _lib.OPT_BWD_CLEARS_DCOLOR, 1)
            try:
                rc = lib.sgs_rasterize_backward(*bargs)
            finally:
                lib.sgs_stream_set_option(sp, _lib.OPT_BWD_CLEARS_DCOLOR, -1 if prev_clear == 0x7fffffff else prev_clear)
            _lib.check(rc, "rasterize_gaussians_backward failed")
    return (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales,
            dL_drotations)


def render_partial(means3D, colors, opacity, scales, rotations, viewmatrix, projmatrix, tan_fovx, tan_fovy,
                   image_height, image_width, campos, scale_modifier=1.0, pool=None, bands=0):
    """One Gaussian SHARD of a view as a compositing partial (A, T): A = the (C,H,W) feature map rendered with a
    ZERO background, T = the (H,W) transmittance left behind the shard.  Partials of depth-ordered shards combine
    with the "over" operator (sgs_hip.dist.composite_over / render_gaussian_sharded, BASELINE config 5):
    (A1, T1) o (A2, T2) = (A1 + T1 A2, T1 T2).  -> (A, T, radii)
    bands = n > 1 (C % 128 == 0): A is the list of the n image bands of dist.band_rows, each a contiguous (C, rows, W) tensor the kernels wrote in
    place (SGS_OPT_OUT_BANDS) -- what render_gaussian_sharded sends without a staging copy."""
    Cn = colors.size(1)
    bg = torch.zeros(Cn, dtype=torch.float32, device=means3D.device)
    e = torch.Tensor([])
    bands = bands if (Cn % 128 == 0 and bands > 1) else 0
    out = None
    if bands:
        try:
            out = rasterize_forward(bg, means3D, colors, opacity, scales, rotations, scale_modifier, e, viewmatrix, projmatrix, tan_fovx,
                                    tan_fovy, image_height, image_width, e, 0, campos, False, False, Cn, False, pool=pool, out_bands=bands)
        except RuntimeError as ex:   # the active blend variant has no band-major output (exact / two-term sweeps, SGS_DEFAULT_SWEEP=14):
            if "SGS_OPT_OUT_BANDS" not in str(ex):   # row-major map, cut into the same bands (one staging copy each, as before round 5)
                raise
    if out is None:
        out = rasterize_forward(bg, means3D, colors, opacity, scales, rotations, scale_modifier, e, viewmatrix, projmatrix, tan_fovx,
                                tan_fovy, image_height, image_width, e, 0, campos, False, False, Cn, False, pool=pool)
        if bands:
            from .dist import band_rows
            out = (out[0], [out[1][:, lo:hi].contiguous() for lo, hi in (band_rows(image_height, b, bands) for b in range(bands))]) + tuple(out[2:])
    _, color, radii, _, _, img, _ = out
    if means3D.size(0) == 0:
        T = torch.ones(image_height, image_width, dtype=torch.float32, device=means3D.device)
    else:
        T = image_views(img, image_width, image_height)["final_T"].clone()
    return color, T, radii


def mark_visible(means3D, viewmatrix, projmatrix):
    """markVisible (CR/rasterize_points.cu:204-223)."""
    lib = _lib.load()
    if not means3D.is_cuda:
        raise RuntimeError("means3D must be a GPU tensor (there is no CPU path)")
    dev = means3D.device
    P = means3D.size(0)
    with torch.cuda.device(dev):
        present = torch.zeros(P, dtype=torch.bool, device=dev)
        if P != 0:
            m_ptr, m = _ptr(means3D, "means3D", dev)
            v_ptr, v = _ptr(viewmatrix, "viewmatrix", dev)
            pr_ptr, pr = _ptr(projmatrix, "projmatrix", dev)
            rc = lib.sgs_mark_visible(P, m_ptr, v_ptr, pr_ptr, present.data_ptr(), _stream_ptr(dev))
            _lib.check(rc, "mark_visible failed")
    return present


def dist2(points):
    """distCUDA2 (SK/spatial.cu:15-26): (P,3) f32 -> (P,) f32."""
    lib = _lib.load()
    if not points.is_cuda:
        raise RuntimeError("points must be a GPU tensor (there is no CPU path)")
    dev = points.device
    P = points.size(0)
    with torch.cuda.device(dev):
        means = torch.zeros(P, dtype=torch.float32, device=dev)
        if P != 0:
            ptr, pts = _ptr(points, "points", dev)
            bufs = _Buffers(dev)
            rc = lib.sgs_knn_mean_dist2(P, ptr, means.data_ptr(), *bufs.callback("s"), _stream_ptr(dev))
            bufs.release()
            _lib.check(rc, "distCUDA2 failed")
    return means


# ---- introspection used by the parity tests (not part of the reference's interface) --------
def geometry_views(geomBuffer, P):
    """Typed views into the opaque geometry buffer."""
    lay = _lib.GeometryLayout()
    _lib.check(_lib.load().sgs_geometry_layout_of(int(P), C.byref(lay)), "layout")
    base = _aligned(geomBuffer)

    def v(off, nbytes, dtype, shape):
        return base[off:off + nbytes].view(dtype).view(*shape)
    return dict(
        depths=v(lay.depths, 4 * P, torch.float32, (P,)),
        clamped=v(lay.clamped, 3 * P, torch.uint8, (P, 3)),
        means2D=v(lay.means2D, 8 * P, torch.float32, (P, 2)),
        cov3D=v(lay.cov3D, 24 * P, torch.float32, (P, 6)),
        conic_opacity=v(lay.conic_opacity, 16 * P, torch.float32, (P, 4)),
        rgb=v(lay.rgb, 12 * P, torch.float32, (P, 3)),
        tiles_touched=v(lay.tiles_touched, 4 * P, torch.int32, (P,)),
        point_offsets=v(lay.point_offsets, 4 * P, torch.int32, (P,)))


def binning_views(binningBuffer, L, geomBuffer=None, P=None, imgBuffer=None, width=None, height=None):
    """Typed views into the opaque binning buffer.  In binning modes 0 and 2 the sorted 64-bit keys
    are only materialised here (pass geomBuffer, P, imgBuffer, width, height); unsorted 64-bit keys
    exist only in mode 1."""
    if geomBuffer is not None and L > 0:
        with torch.cuda.device(binningBuffer.device):
            _lib.check(_lib.load().sgs_debug_sorted_keys(int(P), int(L), int(width), int(height),
                                                         geomBuffer.data_ptr(), binningBuffer.data_ptr(),
                                                         imgBuffer.data_ptr(),
                                                         _stream_ptr(binningBuffer.device)), "keys")
    lay = _lib.BinningLayout()
    _lib.check(_lib.load().sgs_binning_layout_of(int(L), C.byref(lay)), "layout")
    base = _aligned(binningBuffer)

    def v(off, nbytes, dtype):
        return base[off:off + nbytes].view(dtype)
    return dict(
        keys_unsorted=v(lay.keys_unsorted, 8 * L, torch.int64),
        vals_unsorted=v(lay.vals_unsorted, 4 * L, torch.int32),
        keys_sorted=v(lay.keys_sorted, 8 * L, torch.int64),
        point_list=v(lay.point_list, 4 * L, torch.int32))


def image_views(imgBuffer, W, H):
    lay = _lib.ImageLayout()
    _lib.check(_lib.load().sgs_image_layout_of(int(W), int(H), C.byref(lay)), "layout")
    base = _aligned(imgBuffer)
    tiles = ((W + 15) // 16) * ((H + 15) // 16)

    def v(off, nbytes, dtype, shape):
        return base[off:off + nbytes].view(dtype).view(*shape)
    return dict(
        final_T=v(lay.accum_alpha, 4 * W * H, torch.float32, (H, W)),
        n_contrib=v(lay.n_contrib, 4 * W * H, torch.int32, (H, W)),
        ranges=v(lay.ranges, 8 * tiles, torch.int32, (tiles, 2)))


def _aligned(buf):
    """The library carves from the 128-B aligned address inside the caller's chunk."""
    off = (-buf.data_ptr()) % 128
    return buf[off:]


def sort_bits(W, H):
    return int(_lib.load().sgs_sort_bits(int(W), int(H)))


def debug_expf(x):
    lib = _lib.load()
    x = x.contiguous()
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(lib.sgs_debug_expf(x.numel(), x.data_ptr(), out.data_ptr(),
                                      _stream_ptr(x.device)), "debug_expf")
    return out


def set_stream_option(option, value, device=None):
    """Override one tuning option (_lib.OPT_*) for the CURRENT stream of `device` only; value < 0 removes the
    override.  The set_* functions below set the process-wide defaults."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(device):
        return int(_lib.load().sgs_stream_set_option(_stream_ptr(device), int(option), int(value)))


def stream_stat(stat, device=None):
    """Counter (_lib.STAT_*) of the current stream's context: work-list capacity, overflow counts, forwards."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    out = C.c_uint64(0)
    with torch.cuda.device(device):
        _lib.check(_lib.load().sgs_stream_get_stat(_stream_ptr(device), int(stat), C.byref(out)), "stat")
    return int(out.value)


def release_stream(device=None):
    """Free the library's context of the current stream (pinned feedback words, events)."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(device):
        return int(_lib.load().sgs_stream_release(_stream_ptr(device)))


def x16_cu_ownership():
    """Bit 0: the forward's x16 ping-pong sweeps own their compute unit on the current device (they run), bit 1: the fused backward does
    (include/sgs_raster.h sgs_x16_cu_ownership).  3 on a healthy build; anything else means the x8 / fp32-product forms run instead."""
    return int(_lib.load().sgs_x16_cu_ownership())


class PartitionedStreams:
    """`slots` view slots over a chip cut in two compute-unit partitions (include/sgs_raster.h "compute-unit partitions"): slot i owns a
    BLEND stream confined to CUs [front_cus, n) and a FRONT stream confined to CUs [0, front_cus); forwards issued on streams[i] run their
    front end (preprocess -> depth sort -> span partitions) on the small partition and their blend on the large one, so the latency-bound
    front-end kernels of one view never hold the 8-wave sweep workgroups of another off their compute units.  streams[i] is a
    torch.cuda.ExternalStream: use it with torch.cuda.stream(...) like any other.  close() destroys the HIP streams."""

    def __init__(self, device, front_cus, slots, blend_everywhere=False):
        """blend_everywhere: the blend streams are not confined (all n CUs); only the front ends are kept on [0, front_cus)."""
        lib = _lib.load()
        self.device = torch.device(device)
        self.streams, self.front, self._raw = [], [], []
        with torch.cuda.device(self.device):
            n = _lib.check(lib.sgs_device_cu_count(), "cu count")
            if not 0 < front_cus < n:
                raise ValueError(f"front_cus must be in (0, {n})")
            self.cu_count, self.front_cus = n, int(front_cus)
            for _ in range(int(slots)):
                b, f = C.c_void_p(), C.c_void_p()
                if blend_everywhere:
                    _lib.check(lib.sgs_stream_create_cu_range(0, n, C.byref(b)), "blend stream")
                else:
                    _lib.check(lib.sgs_stream_create_cu_range(self.front_cus, n - self.front_cus, C.byref(b)), "blend stream")
                _lib.check(lib.sgs_stream_create_cu_range(0, self.front_cus, C.byref(f)), "front stream")
                _lib.check(lib.sgs_stream_set_front(b, f), "set front")
                self._raw.append((b, f))
                self.streams.append(torch.cuda.ExternalStream(b.value, device=self.device))
                self.front.append(torch.cuda.ExternalStream(f.value, device=self.device))

    def close(self):
        lib = _lib.load()
        with torch.cuda.device(self.device):
            torch.cuda.synchronize(self.device)
            for b, f in self._raw:
                lib.sgs_stream_destroy(b)
                lib.sgs_stream_destroy(f)
        self._raw, self.streams, self.front = [], [], []


def set_blend_exact(exact):
    """Choose the C >= 128 forward blend arithmetic.  False (default): split-bf16 MFMA row sweep,
    feature map within 5e-5 of the absolute composite (north star: 1e-4).  True: fp32 MFMA,
    bit-identical to the reference contract.  Integer state is bit-exact either way.
    Also selectable with SGS_BLEND_EXACT=1 in the environment."""
    return set_blend_variant(15 if exact else 0)


def set_blend_variant(v):
    return int(_lib.load().sgs_set_blend_variant(int(v)))


def set_stage_timing(on):
    return int(_lib.load().sgs_set_stage_timing(int(on)))


def get_stage_ms():
    arr = (C.c_float * 7)()
    _lib.load().sgs_get_stage_ms(arr)
    return [float(a) for a in arr]


def set_binning_mode(mode):
    """0 = depth-presorted emission (default); 1 = reference order of operations (also makes
    point_offsets and the unsorted key/value arrays follow the reference's emission order)."""
    return int(_lib.load().sgs_set_binning_mode(int(mode)))


def build_flags():
    """Optional parts of the loaded library: 1 = make FUSED=1, 2 = make X16=1, 4 = make EXPERIMENTS=1 (include/sgs_raster.h)."""
    return int(_lib.load().sgs_build_flags())


def set_backward_mode(mode):
    """0 = default (C >= 32, C % 32 == 0: the channel work of the backward blend as MFMA products over the
    forward's work list); 1 = always the per-chunk kernel; 2 = as 0 with an undersized arena (tests)."""
    return int(_lib.load().sgs_set_backward_mode(int(mode)))


if os.environ.get("SGS_BLEND_EXACT", "0") not in ("", "0"):
    set_blend_exact(True)

"""Python-side mirror of the reference's rasteriser packages.

One implementation serves both drop-in packages:
  channel_rasterization  (CR/channel_rasterization/__init__.py: N-channel, returns (color, radii))
  rgbd_rasterization     (RR/rgbd_rasterization/__init__.py: C=3, returns (color, radii, depth))
Names, argument order/meaning and error behaviour follow the reference so that
model/renderer.py (render / render_chn) runs unchanged; the work is done by libsgs_hip.so
through sgs_hip.raster.

Deliberate differences (DESIGN.md "Python boundary"):
  * debug=True keeps the reference's contract (device-side check after every stage, input
    snapshot written when a call fails) but does NOT copy every argument to the host before
    every call -- the reference's eager `cpu_deep_copy_tuple` moves the whole (P,C) feature
    table over PCIe per frame (2 GB at 1M x 512); here the snapshot is taken only on failure;
  * backward for num_channels != 3 exists (the reference's is compiled for 3 channels only).
"""
from typing import NamedTuple

import torch
import torch.nn as nn

import os

from . import raster

# no_grad / no input requires grad: enqueue the whole frame before waiting for num_rendered (see _RasterizeGaussians.forward).
# SGS_SPECULATIVE_COUNT=0 restores the wait in the middle of the frame.
SPECULATIVE_COUNT = os.environ.get("SGS_SPECULATIVE_COUNT", "1") not in ("", "0")


class ChannelRasterizationSettings(NamedTuple):
    """Field-for-field the reference's chn GaussianRasterizationSettings
    (CR/channel_rasterization/__init__.py:216-229)."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    num_channels: int


class RgbdRasterizationSettings(NamedTuple):
    """RR/rgbd_rasterization/__init__.py:159-171 (no num_channels: always 3)."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _snapshot(args, path):
    """Input snapshot for a failed call (the reference writes one BEFORE every debug call).  After a real
    device fault the copies to the host fail too: never let that mask the original exception."""
    try:
        host = tuple(a.detach().cpu().clone() if isinstance(a, torch.Tensor) else a for a in args)
        torch.save(host, path)
        return True
    except Exception:   # noqa: BLE001
        return False


def _make_function(with_depth):
    """Builds the autograd bridge (CR/channel_rasterization/__init__.py:47-213; RR/rgbd_rasterization/__init__.py:40-156)."""

    class _RasterizeGaussians(torch.autograd.Function):
        @staticmethod
        def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                    cov3Ds_precomp, raster_settings, track=True):
            s = raster_settings
            channels = 3 if with_depth else s.num_channels
            call = (s.bg, means3D, colors_precomp, opacities, scales, rotations, s.scale_modifier,
                    cov3Ds_precomp, s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy,
                    s.image_height, s.image_width, sh, s.sh_degree, s.campos, s.prefiltered,
                    s.debug, channels)
            # nothing to backpropagate (torch.no_grad, or no input requires grad): the state buffers need
            # not outlive the call, keep them in the resident inference pool.  `track` is decided by the
            # caller of .apply(): ctx.needs_input_grad mirrors the inputs' requires_grad even under no_grad
            # (fusion.py / eval_segmentation.py pass nn.Parameters under torch.no_grad()).
            pool = None if track else raster.INFERENCE_POOL
            try:
                if not track and SPECULATIVE_COUNT:
                    # inference: the host still learns num_rendered before the call returns (the reference's blocking
                    # read-back, rasterizer_impl.cu:283), but the GPU is not left idle while it does: the frame is
                    # enqueued in full against the stream's capacity guess, then the host waits for the counts only
                    # (a frame that outgrew the guess is rendered again; bit-identical results).
                    # Frames that will be differentiated keep the classic order: measured, a training step gains nothing
                    # from it (the optimiser's kernels keep the queue full) and the state buffers would be 1.25 x larger.
                    (num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer, depth) = \
                        raster.rasterize_forward_deferred(*call, want_depth=with_depth, pool=pool).result()
                else:
                    (num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer,
                     depth) = raster.rasterize_forward(*call, want_depth=with_depth, pool=pool)
            except Exception:
                if s.debug and _snapshot(call, "snapshot_fw.dump"):
                    print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise
            ctx.raster_settings = s
            ctx.num_rendered = num_rendered
            ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii,
                                  sh, geomBuffer, binningBuffer, imgBuffer)
            ctx.mark_non_differentiable(radii)
            if with_depth:
                ctx.mark_non_differentiable(depth)   # depth is not differentiable (RR/README.md)
                return color, radii, depth
            return color, radii

        @staticmethod
        def backward(ctx, grad_out_color, *_unused):
            s = ctx.raster_settings
            (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
             binningBuffer, imgBuffer) = ctx.saved_tensors
            call = (s.bg, means3D, radii, colors_precomp, scales, rotations, s.scale_modifier,
                    cov3Ds_precomp, s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy,
                    grad_out_color, sh, s.sh_degree, s.campos, geomBuffer, ctx.num_rendered,
                    binningBuffer, imgBuffer, s.debug)
            try:
                (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D,
                 grad_cov3Ds_precomp, grad_sh, grad_scales,
                 grad_rotations) = raster.rasterize_backward(*call)
            except Exception:
                if s.debug and _snapshot(call, "snapshot_bw.dump"):
                    print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise
            # one gradient per forward input, in input order; None for the settings
            return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities,
                    grad_scales, grad_rotations, grad_cov3Ds_precomp, None, None)

    return _RasterizeGaussians


_ChnFunction = _make_function(with_depth=False)
_RgbdFunction = _make_function(with_depth=True)


def _tracks_grad(*tensors):
    """Will autograd record this call?  (Decided before .apply(): inside forward() grad mode is off.)"""
    return torch.is_grad_enabled() and any(
        isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


def rasterize_gaussians_chn(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                            cov3Ds_precomp, raster_settings):
    track = _tracks_grad(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp)
    return _ChnFunction.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                              cov3Ds_precomp, raster_settings, track)


def rasterize_gaussians_rgbd(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                             cov3Ds_precomp, raster_settings):
    track = _tracks_grad(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp)
    return _RgbdFunction.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                               cov3Ds_precomp, raster_settings, track)


class _RasterizerBase(nn.Module):
    _rasterize = None

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """Boolean mask of points with view-space z > 0.2 (CR/channel_rasterization/__init__.py:237-243)."""
        with torch.no_grad():
            s = self.raster_settings
            return raster.mark_visible(positions, s.viewmatrix, s.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None,
                rotations=None, cov3D_precomp=None):
        # argument contract of the reference (CR/channel_rasterization/__init__.py:258-264), same messages
        if (shs is None) == (colors_precomp is None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        has_sr = scales is not None or rotations is not None
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                has_sr and cov3D_precomp is not None):
            raise Exception(
                "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        empty = torch.Tensor([])   # absent optionals travel as empty CPU tensors
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        return type(self)._rasterize(means3D, means2D, shs, colors_precomp, opacities, scales,
                                     rotations, cov3D_precomp, self.raster_settings)


class ChannelRasterizer(_RasterizerBase):
    _rasterize = staticmethod(rasterize_gaussians_chn)


class RgbdRasterizer(_RasterizerBase):
    _rasterize = staticmethod(rasterize_gaussians_rgbd)

"""Drop-in `rgbd_rasterization` package (reference: submodules/rgbd-rasterization).

    from rgbd_rasterization import GaussianRasterizationSettings, GaussianRasterizer   # model/renderer.py:14
    color, radii, depth = GaussianRasterizer(raster_settings=...)(means3D=..., ...)

RGB (3 channels) plus the median depth map; forward and backward.  Backed by libsgs_hip.so.
"""
from sgs_hip.api import RgbdRasterizationSettings as GaussianRasterizationSettings
from sgs_hip.api import RgbdRasterizer as GaussianRasterizer
from sgs_hip.api import rasterize_gaussians_rgbd as rasterize_gaussians
from . import _C

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "_C"]

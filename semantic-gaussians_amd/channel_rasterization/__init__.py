"""Drop-in `channel_rasterization` package (reference: submodules/channel-rasterization).

    import channel_rasterization as chn_rasterize          # model/renderer.py:15
    chn_rasterize.GaussianRasterizationSettings(..., num_channels=C)
    chn_rasterize.GaussianRasterizer(raster_settings=...)(means3D=..., ...) -> (color, radii)

Backed by libsgs_hip.so (hand-written HIP for gfx950); implementation in sgs_hip.api.
"""
from sgs_hip.api import ChannelRasterizationSettings as GaussianRasterizationSettings
from sgs_hip.api import ChannelRasterizer as GaussianRasterizer
from sgs_hip.api import rasterize_gaussians_chn as rasterize_gaussians
from . import _C

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "_C"]

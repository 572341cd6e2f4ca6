"""Stand-in for the reference's pybind module `channel_rasterization._C` (CR/ext.cpp:15-20):
same three entry points, same positional signatures (CR/rasterize_points.h:18-68)."""
from sgs_hip import raster as _r


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier,
                        cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height,
                        image_width, sh, degree, campos, prefiltered, debug, num_channels):
    out = _r.rasterize_forward(background, means3D, colors, opacity, scales, rotations,
                               scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx,
                               tan_fovy, image_height, image_width, sh, degree, campos,
                               prefiltered, debug, num_channels, want_depth=False)
    return out[:6]


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations,
                                 scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx,
                                 tan_fovy, dL_dout_color, sh, degree, campos, geomBuffer, R,
                                 binningBuffer, imageBuffer, debug):
    return _r.rasterize_backward(background, means3D, radii, colors, scales, rotations,
                                 scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx,
                                 tan_fovy, dL_dout_color, sh, degree, campos, geomBuffer, R,
                                 binningBuffer, imageBuffer, debug)


mark_visible = _r.mark_visible

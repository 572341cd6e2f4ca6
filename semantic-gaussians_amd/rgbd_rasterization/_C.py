"""Stand-in for `rgbd_rasterization._C` (RR/ext.cpp:15-19; RR/rasterize_points.h:18-65):
no debug / num_channels arguments, forward additionally returns the (1,H,W) depth map."""
from sgs_hip import raster as _r


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier,
                        cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height,
                        image_width, sh, degree, campos, prefiltered):
    return _r.rasterize_forward(background, means3D, colors, opacity, scales, rotations,
                                scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx,
                                tan_fovy, image_height, image_width, sh, degree, campos,
                                prefiltered, False, 3, want_depth=True)


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations,
                                 scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx,
                                 tan_fovy, dL_dout_color, sh, degree, campos, geomBuffer, R,
                                 binningBuffer, imageBuffer):
    return _r.rasterize_backward(background, means3D, radii, colors, scales, rotations,
                                 scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx,
                                 tan_fovy, dL_dout_color, sh, degree, campos, geomBuffer, R,
                                 binningBuffer, imageBuffer, False)


mark_visible = _r.mark_visible

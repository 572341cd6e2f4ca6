"""N1: the consumer of the feature map, fused algebraically.

The reference renders the (C,H,W) feature map, normalises every pixel, takes similarities with the
text embeddings and an argmax (eval_segmentation.py:155-157, 255-257):

    rendering = rendering / (rendering.norm(dim=0, keepdim=True) + 1e-8)
    sim = torch.einsum("cq,qhw->chw", text_features, rendering)        # (n_cls, H, W)
    label = sim[1:].argmax(dim=0)

Alpha compositing is linear in the features, so the similarity of a composited feature is the
composite of the per-Gaussian similarities:

    sum_q text[c,q] * (sum_k F[k,q] w[k,px] + bg[q] T[px]) = sum_k (F @ text.T)[k,c] w[k,px] + (text @ bg)[c] T[px]

i.e. rendering the n_cls-channel table  F @ text.T  with background  text @ bg  yields the UNNORMALISED
similarities directly -- C = n_cls (20-200) instead of 512-768 channels, no 2.57 GB feature map written,
normalised, and read back.  The per-pixel normalisation is a positive scalar, so the argmax -- the label --
is unchanged by it; only callers that need the normalised similarity VALUES still need the norm of the
full feature vector.  That norm is one (H,W) plane: `render_norm2` has the blend add each 128-channel group's
sum of squares into it with the feature-map stores suppressed (SGS_OPT_NORM_PLANE, include/sgs_raster.h), so
`render_similarity(..., normalised=True)` = logits / (sqrt(norm2) + 1e-8) never materialises the (C,H,W) map either.
Opt-in: the drop-in rasteriser API is untouched.
"""
import torch

from . import api, raster


def project_features(features, text_features):
    """(P,C) Gaussian features x (n_cls,C) text embeddings -> (P,n_cls) per-Gaussian similarities.
    Once per (scene, text set); one GEMM."""
    return (features @ text_features.t()).contiguous()


def render_logits(raster_settings, means3D, opacities, scales, rotations, projected, text_features):
    """Unnormalised similarities (n_cls,H,W) of one view.  `raster_settings` is the
    GaussianRasterizationSettings the caller would use for the feature render (its bg is the C-dim
    background; num_channels is replaced); `projected` = project_features(features, text_features)."""
    n_cls = projected.shape[1]
    bg = raster_settings.bg
    bg_c = bg[:text_features.shape[1]].to(text_features.dtype)
    s = raster_settings._replace(num_channels=n_cls, bg=(text_features @ bg_c).contiguous())
    rast = api.ChannelRasterizer(s)
    logits, radii = rast(means3D=means3D, means2D=torch.zeros_like(means3D), opacities=opacities,
                         colors_precomp=projected, scales=scales, rotations=rotations)
    return logits, radii


def labels_from_logits(logits, skip_first=True):
    """The reference's `sim[1:].argmax(dim=0)` (class 0 is the 'other' prompt); add 1 for its label ids."""
    return (logits[1:] if skip_first else logits).argmax(dim=0)


def render_norm2(raster_settings, means3D, opacities, scales, rotations, features):
    """(H,W) plane  sum_c render[c]^2  of the C-channel feature render, without writing the render
    (C % 128 == 0; inference only -- no autograd)."""
    s = raster_settings
    empty = torch.Tensor([])
    with torch.no_grad():
        out = raster.rasterize_forward(s.bg, means3D, features, opacities, scales, rotations, s.scale_modifier,
                                       empty, s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, s.image_height,
                                       s.image_width, empty, s.sh_degree, s.campos, s.prefiltered, s.debug,
                                       s.num_channels, want_depth=False, pool=raster.INFERENCE_POOL, norm_plane=True)
    return out[1]


def render_similarity(raster_settings, means3D, opacities, scales, rotations, features, text_features,
                      normalised=False, projected=None):
    """(n_cls,H,W) similarities.  normalised=False: the projected render (fast path, argmax-equivalent).
    normalised=True: the reference's values (eval_segmentation.py:155-156): the projected render divided by the
    per-pixel norm of the full feature vector (render_norm2; C % 128 != 0 falls back to rendering the map).
    `projected` = project_features(features, text_features) if the caller keeps it across views."""
    if projected is None:
        projected = project_features(features, text_features)
    if not normalised:
        return render_logits(raster_settings, means3D, opacities, scales, rotations, projected, text_features)[0]
    if raster_settings.num_channels % 128 == 0:
        logits = render_logits(raster_settings, means3D, opacities, scales, rotations, projected, text_features)[0]
        norm2 = render_norm2(raster_settings, means3D, opacities, scales, rotations, features)
        return logits / (norm2.sqrt() + 1e-8)
    rast = api.ChannelRasterizer(raster_settings)
    rendering, _ = rast(means3D=means3D, means2D=torch.zeros_like(means3D), opacities=opacities,
                        colors_precomp=features, scales=scales, rotations=rotations)
    rendering = rendering / (rendering.norm(dim=0, keepdim=True) + 1e-8)
    return torch.einsum("cq,qhw->chw", text_features, rendering)

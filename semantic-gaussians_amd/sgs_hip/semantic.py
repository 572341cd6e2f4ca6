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
full feature vector (`render_similarity(..., normalised=True)` renders the feature map for that).
Opt-in: the drop-in rasteriser API is untouched.
"""
import torch

from . import api


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


def render_similarity(raster_settings, means3D, opacities, scales, rotations, features, text_features,
                      normalised=False):
    """(n_cls,H,W) similarities.  normalised=False: the projected render (fast path, argmax-equivalent).
    normalised=True: the reference's values -- renders the full feature map for the per-pixel norm."""
    if not normalised:
        return render_logits(raster_settings, means3D, opacities, scales, rotations,
                             project_features(features, text_features), text_features)[0]
    rast = api.ChannelRasterizer(raster_settings)
    rendering, _ = rast(means3D=means3D, means2D=torch.zeros_like(means3D), opacities=opacities,
                        colors_precomp=features, scales=scales, rotations=rotations)
    rendering = rendering / (rendering.norm(dim=0, keepdim=True) + 1e-8)
    return torch.einsum("cq,qhw->chw", text_features, rendering)

"""N1: semantic labels from the projected (n_cls-channel) render vs the reference's consumer formula
(eval_segmentation.py:155-157: normalise the rendered feature map, einsum with the text features, argmax)."""
import numpy as np
import pytest
import torch

from helpers import small_scene, oracle_forward


def _reference_consumer(rendering, text):
    rendering = rendering / (rendering.norm(dim=0, keepdim=True) + 1e-8)
    sim = torch.einsum("cq,qhw->chw", text, rendering)
    return sim, sim[1:].argmax(dim=0)


def _text(n_cls, C, seed):
    g = torch.Generator().manual_seed(seed)
    t = torch.randn(n_cls, C, generator=g)
    return t / t.norm(dim=1, keepdim=True)


def test_projection_commutes_with_compositing_on_the_oracle(orc):
    """CPU: compositing is linear in the features, so rendering F @ text.T with background text @ bg gives
    text @ (the rendered feature map) -- the identity the fast path rests on."""
    scene, cam = small_scene(P=1500, C=24, W=96, H=64, fx=80.0, seed=31)
    scene = scene._replace(bg=torch.linspace(-0.2, 0.3, 24))
    text = _text(7, 24, 1)
    full = torch.from_numpy(oracle_forward(orc, scene, cam)["out"])
    proj = scene._replace(features=(scene.features @ text.t()).contiguous(), bg=(text @ scene.bg).contiguous())
    logits = torch.from_numpy(oracle_forward(orc, proj, cam)["out"])
    want = torch.einsum("cq,qhw->chw", text, full)
    assert float((logits - want).abs().max()) < 1e-5
    _, label_ref = _reference_consumer(full, text)
    agree = (logits[1:].argmax(dim=0) == label_ref).float().mean()
    assert float(agree) > 0.995      # the positive per-pixel normalisation cannot change the argmax; near-ties may flip


@pytest.mark.gpu
def test_render_logits_matches_reference_consumer(orc):
    from sgs_hip import raster, semantic
    import channel_rasterization as cr
    dev = "cuda:0"
    C, n_cls, W, H = 256, 21, 208, 128
    scene, cam = small_scene(P=4000, C=C, W=W, H=H, fx=170.0, seed=41)
    text = _text(n_cls, C, 2)
    s, c, t = scene.to(dev), cam.to(dev), text.to(dev)
    settings = cr.GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=s.bg, scale_modifier=1.0,
        viewmatrix=c.world_view_transform, projmatrix=c.full_proj_transform, sh_degree=0, campos=c.camera_center,
        prefiltered=False, debug=False, num_channels=C)
    with torch.no_grad():
        proj = semantic.project_features(s.features, t)
        logits, radii = semantic.render_logits(settings, s.means3D, s.opacities, s.scales, s.rotations, proj, t)
        sim_fast = semantic.render_similarity(settings, s.means3D, s.opacities, s.scales, s.rotations, s.features, t)
        sim_norm = semantic.render_similarity(settings, s.means3D, s.opacities, s.scales, s.rotations, s.features, t,
                                              normalised=True)
    assert logits.shape == (n_cls, H, W) and torch.equal(logits, sim_fast)
    # against the oracle's feature map pushed through the reference's own formula
    full = torch.from_numpy(oracle_forward(orc, scene, cam)["out"])
    sim_ref, label_ref = _reference_consumer(full, text)
    want = torch.einsum("cq,qhw->chw", text, full)
    assert float((logits.cpu() - want).abs().max()) < 2e-5 * float(want.abs().max() + 1)
    labels = semantic.labels_from_logits(logits).cpu()
    assert float((labels == label_ref).float().mean()) > 0.995
    top2 = torch.topk(sim_ref[1:], 2, dim=0).values
    margin = (top2[0] - top2[1])[labels != label_ref]
    assert margin.numel() == 0 or float(margin.max()) < 1e-4      # only near-ties can differ
    assert float((sim_norm.cpu() - sim_ref).abs().max()) < 1e-4   # the normalised values path
    assert np.array_equal(radii.cpu().numpy(), oracle_forward(orc, scene, cam)["radii"])

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


def _settings(cr, c, s, W, H, C):
    return cr.GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=s.bg, scale_modifier=1.0,
        viewmatrix=c.world_view_transform, projmatrix=c.full_proj_transform, sh_degree=0, campos=c.camera_center,
        prefiltered=False, debug=False, num_channels=C)


@pytest.mark.gpu
@pytest.mark.parametrize("C,W,H,P,exact", [(128, 208, 128, 3000, False), (256, 203, 117, 5000, False),
                                           (512, 330, 90, 2500, True)])
def test_norm_plane_is_the_squared_norm_of_the_render(orc, C, W, H, P, exact):
    """SGS_OPT_NORM_PLANE: the (H,W) plane sum_c render[c]^2 without the render -- vs the oracle's feature map
    (ragged widths / heights, several channel groups adding into the same pixels, a non-zero background)."""
    from sgs_hip import raster, semantic
    import channel_rasterization as cr
    dev = "cuda:0"
    scene, cam = small_scene(P=P, C=C, W=W, H=H, fx=170.0, seed=43 + C)
    scene = scene._replace(bg=torch.linspace(-0.3, 0.4, C))
    s, c = scene.to(dev), cam.to(dev)
    settings = _settings(cr, c, s, W, H, C)
    prev = raster.set_blend_variant(15 if exact else 0)
    try:
        n2 = semantic.render_norm2(settings, s.means3D, s.opacities, s.scales, s.rotations, s.features)
        # the option was consumed: the next forward on the stream renders the map again
        full_gpu, _ = cr.GaussianRasterizer(settings)(means3D=s.means3D, means2D=torch.zeros_like(s.means3D), opacities=s.opacities,
                                                      colors_precomp=s.features, scales=s.scales, rotations=s.rotations)
    finally:
        raster.set_blend_variant(prev)
    assert n2.shape == (H, W) and full_gpu.shape == (C, H, W)
    full = torch.from_numpy(oracle_forward(orc, scene, cam)["out"]).double()
    want = (full * full).sum(dim=0)
    got = n2.cpu().double()
    assert float(((got - want).abs() / (want + 1e-6)).max()) < 2e-5      # fp32 squares and sums of C terms
    mine = (full_gpu.double() ** 2).sum(dim=0).cpu()
    assert float(((got - mine).abs() / (mine + 1e-6)).max()) < 2e-6      # the same render, summed here instead


@pytest.mark.gpu
def test_norm_plane_with_nothing_rendered_and_bad_arguments():
    from sgs_hip import raster, semantic
    import channel_rasterization as cr
    dev = "cuda:0"
    C, W, H = 128, 64, 48
    scene, cam = small_scene(P=200, C=C, W=W, H=H, fx=60.0, seed=5)
    scene = scene._replace(bg=torch.linspace(0.1, 0.9, C))
    s, c = scene.to(dev), cam.to(dev)
    behind = s.means3D.clone()
    behind[:, 2] = -5.0 - behind[:, 2].abs()      # nothing in front of the camera: num_rendered == 0
    far = torch.einsum("ij,nj->ni", c.world_view_transform.t()[:3, :3], behind) + c.world_view_transform.t()[:3, 3]
    if float(far[:, 2].max()) > 0.2:
        behind = -s.means3D
    n2 = semantic.render_norm2(_settings(cr, c, s, W, H, C), behind, s.opacities, s.scales, s.rotations, s.features)
    want = float((scene.bg.double() ** 2).sum())
    vis = raster.mark_visible(behind, c.world_view_transform, c.full_proj_transform)
    if not bool(vis.any()):
        assert torch.allclose(n2.cpu().double(), torch.full((H, W), want, dtype=torch.float64), rtol=1e-6)
    with pytest.raises(RuntimeError):
        semantic.render_norm2(_settings(cr, c, s, W, H, 96), s.means3D, s.opacities, s.scales, s.rotations,
                              s.features[:, :96].contiguous())


@pytest.mark.gpu
def test_norm_plane_through_the_overflow_fallback(orc):
    """A fresh stream's first frame of a dense scene outgrows the initial work-list capacity: the sweep exits and the gated
    single-kernel fallback renders the frame -- in norm mode it must add the squares into the plane, not write a feature map."""
    from sgs_hip import raster, semantic, _lib
    import channel_rasterization as cr
    dev = "cuda:0"
    C, W, H = 128, 784, 32      # 98 tiles with ~900 active entries each: more than the 64k slots a fresh stream starts with
    scene, cam = small_scene(P=120000, C=C, W=W, H=H, fx=600.0, seed=77)
    scene = scene._replace(scales=scene.scales * 3.0, opacities=scene.opacities * 0.02, bg=torch.linspace(-0.2, 0.2, C))
    s, c = scene.to(dev), cam.to(dev)
    full = torch.from_numpy(oracle_forward(orc, scene, cam)["out"]).double()
    want = (full * full).sum(dim=0)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        n2_first = semantic.render_norm2(_settings(cr, c, s, W, H, C), s.means3D, s.opacities, s.scales, s.rotations, s.features)
        n2_second = semantic.render_norm2(_settings(cr, c, s, W, H, C), s.means3D, s.opacities, s.scales, s.rotations, s.features)
        overflows = raster.stream_stat(_lib.STAT_FWD_OVERFLOWS)
    torch.cuda.synchronize()
    assert overflows >= 1      # the first frame really took the fallback (reported when the second frame read the feedback)
    for n2 in (n2_first, n2_second):
        assert float(((n2.cpu().double() - want).abs() / (want + 1e-6)).max()) < 2e-5

"""the rasterisers inside the reference's optimisation loop (train.py:125-185,
model/renderer.py:54-111): activations -> GaussianRasterizer -> loss -> backward -> Adam, with the
densification hooks train.py reads (viewspace_points.grad, radii, visibility_filter).  Functional test: the
loss must go down, the hooks must carry what add_densification_stats consumes
(model/gaussian_model.py:608-612)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(P, C, W, H, fx, seed):
    from helpers import small_scene
    scene, cam = small_scene(P=P, C=C, W=W, H=H, fx=fx, seed=seed)
    return scene.to(DEV), cam.to(DEV)


def _params(s, g, noise):
    """Raw (pre-activation) parameters as GaussianModel keeps them, perturbed."""
    xyz = (s.means3D + noise * 0.02 * torch.randn(s.means3D.shape, generator=g, device=DEV)).requires_grad_(True)
    opacity = torch.logit(s.opacities.clamp(1e-3, 1 - 1e-3)).add(noise * torch.randn(s.opacities.shape, generator=g, device=DEV)).requires_grad_(True)
    scaling = torch.log(s.scales).add(noise * 0.1 * torch.randn(s.scales.shape, generator=g, device=DEV)).requires_grad_(True)
    rotation = (s.rotations + noise * 0.05 * torch.randn(s.rotations.shape, generator=g, device=DEV)).requires_grad_(True)
    return xyz, opacity, scaling, rotation


def _fit(module, C, steps, with_depth):
    s, c = _setup(2500, C, 96, 64, 85.0, seed=31 + C)
    g = torch.Generator(device=DEV).manual_seed(0)
    kw = dict(image_height=64, image_width=96, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=torch.zeros(C, device=DEV),
              scale_modifier=1.0, viewmatrix=c.world_view_transform, projmatrix=c.full_proj_transform, sh_degree=0,
              campos=c.camera_center, prefiltered=False, debug=False)
    if not with_depth:
        kw["num_channels"] = C
    rast = module.GaussianRasterizer(raster_settings=module.GaussianRasterizationSettings(**kw))

    def render(xyz, colors, opacity, scaling, rotation):
        screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=DEV) + 0
        if screenspace_points.requires_grad:   # (not under torch.no_grad(): the target render)
            screenspace_points.retain_grad()
        out = rast(means3D=xyz, means2D=screenspace_points, shs=None, colors_precomp=colors,
                   opacities=torch.sigmoid(opacity), scales=torch.exp(scaling),
                   rotations=torch.nn.functional.normalize(rotation), cov3D_precomp=None)
        return out, screenspace_points

    with torch.no_grad():
        target = render(s.means3D, s.features, torch.logit(s.opacities.clamp(1e-3, 1 - 1e-3)), torch.log(s.scales), s.rotations)[0][0]
    xyz, opacity, scaling, rotation = _params(s, g, noise=1.0)
    colors = (s.features + 0.3 * torch.randn(s.features.shape, generator=g, device=DEV)).requires_grad_(True)
    opt = torch.optim.Adam([{"params": [xyz], "lr": 1e-3}, {"params": [colors], "lr": 2e-2}, {"params": [opacity], "lr": 5e-2},
                            {"params": [scaling], "lr": 5e-3}, {"params": [rotation], "lr": 1e-3}], eps=1e-15)
    xyz_gradient_accum = torch.zeros(xyz.shape[0], 1, device=DEV)
    denom = torch.zeros(xyz.shape[0], 1, device=DEV)
    max_radii2D = torch.zeros(xyz.shape[0], device=DEV)
    losses = []
    for it in range(steps):
        out, viewspace_point_tensor = render(xyz, colors, opacity, scaling, rotation)
        image, radii = out[0], out[1]
        loss = (image - target).abs().mean()
        loss.backward()
        with torch.no_grad():
            visibility_filter = radii > 0
            max_radii2D[visibility_filter] = torch.max(max_radii2D[visibility_filter], radii[visibility_filter].float())
            xyz_gradient_accum[visibility_filter] += torch.norm(viewspace_point_tensor.grad[visibility_filter, :2], dim=-1, keepdim=True)
            denom[visibility_filter] += 1
            opt.step()
            opt.zero_grad()
        losses.append(float(loss.detach()))
    assert radii.dtype == torch.int32 and viewspace_point_tensor.grad.shape == xyz.shape
    assert float(xyz_gradient_accum.sum()) > 0 and int(denom.max()) == steps and float(max_radii2D.max()) > 0
    if with_depth:
        assert out[2].shape == (1, 64, 96) and not out[2].requires_grad
    return losses


def test_rgbd_training_loop_converges():
    import rgbd_rasterization
    losses = _fit(rgbd_rasterization, 3, steps=80, with_depth=True)
    assert losses[-1] < 0.5 * losses[0], (losses[0], losses[-1])


@pytest.mark.parametrize("C", [64, 20])
def test_feature_training_loop_converges(C):
    """N-channel gradients (which the reference's backward cannot produce): C = 64 runs the work-list MFMA
    backward, C = 20 the per-chunk kernel."""
    import channel_rasterization
    losses = _fit(channel_rasterization, C, steps=80, with_depth=False)
    assert losses[-1] < 0.5 * losses[0], (losses[0], losses[-1])


def test_densification_statistics_match_the_oracle(orc):
    """Parity, not property: ONE optimisation step's densification inputs -- what train.py:156-160 and
    add_densification_stats (model/gaussian_model.py:608-612) read -- against the oracle's backward of the same
    loss: viewspace_points.grad[:, :2] (hence xyz_gradient_accum), radii (max_radii2D), visibility_filter
    (denom); and the densify mask `grads >= densify_grad_threshold` derived from them."""
    import numpy as np
    import rgbd_rasterization
    from helpers import small_scene, oracle_forward
    scene, cam = small_scene(P=2500, C=3, W=96, H=64, fx=85.0, seed=34)
    s, c = scene.to(DEV), cam.to(DEV)
    g = torch.Generator().manual_seed(7)
    target = torch.rand(3, 64, 96, generator=g)
    bg = torch.tensor([0.1, 0.2, 0.3])
    kw = dict(image_height=64, image_width=96, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=bg.to(DEV), scale_modifier=1.0,
              viewmatrix=c.world_view_transform, projmatrix=c.full_proj_transform, sh_degree=0, campos=c.camera_center,
              prefiltered=False, debug=False)
    rast = rgbd_rasterization.GaussianRasterizer(raster_settings=rgbd_rasterization.GaussianRasterizationSettings(**kw))
    xyz = s.means3D.clone().requires_grad_(True)
    screenspace_points = torch.zeros_like(xyz, requires_grad=True) + 0
    screenspace_points.retain_grad()
    image, radii, depth = rast(means3D=xyz, means2D=screenspace_points, shs=None, colors_precomp=s.features,
                               opacities=s.opacities, scales=s.scales, rotations=s.rotations, cov3D_precomp=None)
    loss = ((image - target.to(DEV)) ** 2).mean()   # (smooth: an L1 residual near 0 could flip sign between renders)
    loss.backward()
    visibility_filter = radii > 0
    accum = torch.norm(screenspace_points.grad[visibility_filter, :2], dim=-1)

    fw = oracle_forward(orc, scene, cam, bg=bg.numpy())
    dL = (2.0 * (fw["out"] - target.numpy()) / fw["out"].size).astype(np.float32)    # d mean (x - t)^2 / dx
    gb = orc.backward(fw, dL, scene.means3D.numpy(), cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(),
                      cam.camera_center.numpy(), 96, 64, cam.tanfovx, cam.tanfovy, bg.numpy(),
                      scales=scene.scales.numpy(), rotations=scene.rotations.numpy())
    vis_o = fw["radii"] > 0
    assert np.array_equal(radii.cpu().numpy(), fw["radii"]) and np.array_equal(visibility_filter.cpu().numpy(), vis_o)
    want = np.linalg.norm(gb["dL_dmean2D"][vis_o, :2], axis=-1)
    got = accum.cpu().numpy()
    assert np.abs(got - want).max() <= 1e-4 * want.max()
    assert np.abs(xyz.grad.cpu().numpy() - gb["dL_dmeans3D"]).max() <= 1e-4 * np.abs(gb["dL_dmeans3D"]).max()
    # the densification decision (densify_and_prune: grads >= threshold) is the same set, away from the threshold
    thr = float(np.median(want))
    away = np.abs(want - thr) > 1e-3 * thr
    assert np.array_equal((got >= thr)[away], (want >= thr)[away])

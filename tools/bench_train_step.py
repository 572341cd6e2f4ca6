"""N2 measurement: one optimisation step as train.py runs it (render -> L1 -> backward -> Adam) at cfg2
(500k Gaussians, RGB-D rasteriser, 968x1296), and the same with a 512-channel feature target at cfg3's size
(distillation-style; the reference's backward cannot run this)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-gaussians_amd"))
import torch
import rgbd_rasterization as rr
import channel_rasterization as cr
from sgs_hip.synthetic import CONFIGS, make_scene
from sgs_hip.camera import pinhole

dev = "cuda:0"
for name, C in (("cfg2", 3), ("cfg3", 512)):
    P, _, W, H, fx = CONFIGS[name]
    s = make_scene(P, C, W, H, fx, seed=0).to(dev)
    c = pinhole(W, H, fx).to(dev)
    mod = rr if C == 3 else cr
    kw = dict(image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=s.bg, scale_modifier=1.0,
              viewmatrix=c.world_view_transform, projmatrix=c.full_proj_transform, sh_degree=0, campos=c.camera_center,
              prefiltered=False, debug=False)
    if C != 3:
        kw["num_channels"] = C
    rast = mod.GaussianRasterizer(mod.GaussianRasterizationSettings(**kw))
    xyz = s.means3D.clone().requires_grad_(True)
    colors = s.features.clone().requires_grad_(True)
    opacity = torch.logit(s.opacities.clamp(1e-3, 1 - 1e-3)).requires_grad_(True)
    scaling = torch.log(s.scales).requires_grad_(True)
    rotation = s.rotations.clone().requires_grad_(True)
    opt = torch.optim.Adam([xyz, colors, opacity, scaling, rotation], lr=1e-4, eps=1e-15)
    target = torch.rand(C, H, W, device=dev)

    def step():
        m2d = torch.zeros_like(xyz, requires_grad=True) + 0
        out = rast(means3D=xyz, means2D=m2d, shs=None, colors_precomp=colors, opacities=torch.sigmoid(opacity),
                   scales=torch.exp(scaling), rotations=torch.nn.functional.normalize(rotation), cov3D_precomp=None)
        loss = (out[0] - target).abs().mean()
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)

    N = 20 if C == 3 else 5
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        step()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / N
    print(f"{name} P={P} C={C} {W}x{H}: render + L1 + backward + Adam = {t * 1e3:.2f} ms per iteration ({1 / t:.0f} it/s)")

"""Forward + backward timing of the RGB-D path (cfg2: 500k Gaussians, C=3, 968x1296) and of an
N-channel backward (C=32).  Development measurement; the headline metric is the forward (bench.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-gaussians_amd"))
import torch
import rgbd_rasterization as rr
import channel_rasterization as cr
from sgs_hip.synthetic import CONFIGS, make_scene
from sgs_hip.camera import pinhole

dev = "cuda:0"
from sgs_hip import raster
for name, C, bmode in (("cfg2", 3, 0), ("cfg2", 32, 0), ("cfg3", 512, 0), ("cfg3", 512, 1)):
    P, _, W, H, fx = CONFIGS[name]
    raster.set_backward_mode(bmode)   # 0 = work-list MFMA backward for C >= 128, 1 = per-chunk kernel
    N = 10 if C <= 32 else 3
    scene = make_scene(P, C, W, H, fx, seed=0).to(dev)
    cam = pinhole(W, H, fx).to(dev)
    mod = rr if C == 3 else cr
    kw = dict(image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=scene.bg, scale_modifier=1.0,
              viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=0,
              campos=cam.camera_center, prefiltered=False, debug=False)
    if C != 3:
        kw["num_channels"] = C
    rast = mod.GaussianRasterizer(mod.GaussianRasterizationSettings(**kw))
    leaves = [t.clone().requires_grad_(True) for t in (scene.means3D, scene.opacities, scene.features, scene.scales, scene.rotations)]
    m2d = torch.zeros_like(scene.means3D, requires_grad=True)

    def fwd():
        return rast(means3D=leaves[0], means2D=m2d, opacities=leaves[1], colors_precomp=leaves[2], scales=leaves[3], rotations=leaves[4])

    for _ in range(2):
        out = fwd()
        out[0].sum().backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        out = fwd()
    torch.cuda.synchronize()
    tf = (time.perf_counter() - t0) / N
    t0 = time.perf_counter()
    for _ in range(N):
        out = fwd()
        out[0].sum().backward()
    torch.cuda.synchronize()
    tfb = (time.perf_counter() - t0) / N
    print(f"{name} P={P} C={C} {W}x{H} backward_mode={bmode}: forward {tf * 1e3:.2f} ms, forward+backward {tfb * 1e3:.2f} ms")

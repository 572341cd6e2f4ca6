"""cfg3 (1M Gaussians, C = 512, 968x1296) forward + backward through the drop-in module: median / min of N iterations, and of
the backward alone (events around .backward()).  Development measurement (A/B runs: SGS_NO_TILE_ORDER=1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-gaussians_amd"))
import torch
import channel_rasterization as cr
from sgs_hip.synthetic import CONFIGS, make_scene
from sgs_hip.camera import pinhole

dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
P, _, W, H, fx = CONFIGS["cfg3"]
C = 512
scene = make_scene(P, C, W, H, fx, seed=0).to(dev)
cam = pinhole(W, H, fx).to(dev)
kw = dict(image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=scene.bg, scale_modifier=1.0,
          viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=0,
          campos=cam.camera_center, prefiltered=False, debug=False, num_channels=C)
rast = cr.GaussianRasterizer(cr.GaussianRasterizationSettings(**kw))
leaves = [t.clone().requires_grad_(True) for t in (scene.means3D, scene.opacities, scene.features, scene.scales, scene.rotations)]
m2d = torch.zeros_like(scene.means3D, requires_grad=True)
g = torch.randn(C, H, W, device=dev)
tb, tfb = [], []
for it in range(N + 3):
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    for t in leaves + [m2d]:
        t.grad = None
    e0.record()
    out = rast(means3D=leaves[0], means2D=m2d, opacities=leaves[1], colors_precomp=leaves[2], scales=leaves[3], rotations=leaves[4])[0]
    e1.record()
    out.backward(g)
    e2.record()
    torch.cuda.synchronize()
    if it >= 3:
        tb.append(e1.elapsed_time(e2))
        tfb.append(e0.elapsed_time(e2))
    del out
tb.sort(); tfb.sort()
print(f"cfg3 C=512 tile_order={'off' if os.environ.get('SGS_NO_TILE_ORDER') == '1' else 'on'}: backward median {tb[len(tb) // 2]:.3f} min {tb[0]:.3f} ms; "
      f"forward+backward median {tfb[len(tfb) // 2]:.3f} min {tfb[0]:.3f} ms")

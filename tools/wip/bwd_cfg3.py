"""cfg3 forward + backward only (for rocprofv3 --kernel-trace --stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "semantic-gaussians_amd"))
import torch
from sgs_hip import raster
from sgs_hip.synthetic import CONFIGS, make_scene
from sgs_hip.camera import pinhole
dev = "cuda:0"; E = torch.Tensor([])
P, C, W, H, fx = CONFIGS["cfg3"]
s = make_scene(P, C, W, H, fx, seed=0).to(dev); c = pinhole(W, H, fx).to(dev)
dL = torch.randn(C, H, W, device=dev)
if len(sys.argv) > 1: raster.set_backward_mode(int(sys.argv[1]))
def it():
    n, col, rad, g_, b_, i_, _ = raster.rasterize_forward(s.bg, s.means3D, s.features, s.opacities, s.scales, s.rotations, 1.0, E,
        c.world_view_transform, c.full_proj_transform, c.tanfovx, c.tanfovy, H, W, E, 0, c.camera_center, False, False, C, False)
    raster.rasterize_backward(s.bg, s.means3D, rad, s.features, s.scales, s.rotations, 1.0, E, c.world_view_transform,
        c.full_proj_transform, c.tanfovx, c.tanfovy, dL, E, 0, c.camera_center, g_, n, b_, i_, False)
for _ in range(3): it()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
for a, b in ev:
    a.record(); it(); b.record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in ev); print("fwd+bwd ms median", ms[len(ms) // 2], "min", ms[0])

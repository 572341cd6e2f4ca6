"""cfg3 forward + backward N times on one stream (default backward mode: the persistent fused kernel): every iteration's gradients against the first
iteration's -- equal up to the order of the colour / geometry atomics.  python tools/soak_backward.py [N=1000]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-gaussians_amd"))
import torch
from sgs_hip import raster
from sgs_hip.synthetic import CONFIGS, make_scene
from sgs_hip.camera import pinhole

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = "cuda:0"
P, C, W, H, fx = CONFIGS["cfg3"]
s = make_scene(P, C, W, H, fx, seed=0).to(dev)
c = pinhole(W, H, fx).to(dev)
empty = torch.empty(0, device=dev)
dL = torch.randn(C, H, W, device=dev)


def fwd_bwd():
    n, col, rad, g_, b_, i_, _ = raster.rasterize_forward(
        s.bg, s.means3D, s.features, s.opacities, s.scales, s.rotations, 1.0, empty, c.world_view_transform,
        c.full_proj_transform, c.tanfovx, c.tanfovy, H, W, empty, 0, c.camera_center, False, False, C, False)
    return raster.rasterize_backward(s.bg, s.means3D, rad, s.features, s.scales, s.rotations, 1.0, empty,
                                     c.world_view_transform, c.full_proj_transform, c.tanfovx, c.tanfovy, dL,
                                     empty, 0, c.camera_center, g_, n, b_, i_, False)


ref = [g.clone() for g in fwd_bwd() if g.numel()]
scale = [max(float(r.abs().max()), 1e-30) for r in ref]
worst = [0.0] * len(ref)
bad = 0
for k in range(N):
    out = [g for g in fwd_bwd() if g.numel()]
    for j, (g, r) in enumerate(zip(out, ref)):
        d = float((g - r).abs().max()) / scale[j]
        worst[j] = max(worst[j], d)
        if not d < 1e-5:
            bad += 1
torch.cuda.synchronize()
print(f"{N} forward + backward at cfg3: worst |g - g0| / max|g0| per gradient tensor: " + " ".join(f"{w:.1e}" for w in worst) + f"; iterations beyond 1e-5: {bad}")

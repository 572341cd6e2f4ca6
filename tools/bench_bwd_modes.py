"""cfg3 forward + backward through the C-ABI, per backward mode (0 split-bf16 products, 3 fp32 products).
Development measurement: python tools/bench_bwd_modes.py [modes...]   (rocprofv3 --kernel-trace --stats friendly)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-gaussians_amd"))
import torch
from sgs_hip import raster
from sgs_hip.synthetic import CONFIGS, make_scene
from sgs_hip.camera import pinhole

dev = "cuda:0"
modes = [int(a) for a in sys.argv[1:]] or [0, 3]
P, C, W, H, fx = CONFIGS["cfg3"]
s = make_scene(P, C, W, H, fx, seed=0).to(dev)
c = pinhole(W, H, fx).to(dev)
empty = torch.empty(0, device=dev)
dL = torch.randn(C, H, W, device=dev)
ref = None
for mode in modes:
    raster.set_backward_mode(mode)

    def fwd_bwd():
        n, col, rad, g_, b_, i_, _ = raster.rasterize_forward(
            s.bg, s.means3D, s.features, s.opacities, s.scales, s.rotations, 1.0, empty, c.world_view_transform,
            c.full_proj_transform, c.tanfovx, c.tanfovy, H, W, empty, 0, c.camera_center, False, False, C, False)
        return raster.rasterize_backward(s.bg, s.means3D, rad, s.features, s.scales, s.rotations, 1.0, empty,
                                         c.world_view_transform, c.full_proj_transform, c.tanfovx, c.tanfovy, dL,
                                         empty, 0, c.camera_center, g_, n, b_, i_, False)
    for _ in range(3):
        out = fwd_bwd()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fwd_bwd()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    grads = [t for t in out if t.numel()]
    msg = f"backward_mode={mode}: forward+backward median {ts[len(ts) // 2]:.3f} ms (min {ts[0]:.3f})"
    if ref is None:
        ref = [g.clone() for g in grads]
    else:
        msg += "; max|diff| / max|ref| vs first mode: " + " ".join(
            f"{float((g - r).abs().max()) / max(float(r.abs().max()), 1e-30):.1e}" for g, r in zip(grads, ref))
    print(msg, flush=True)
raster.set_backward_mode(0)

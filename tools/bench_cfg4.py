"""cfg4 (5M Gaussians x 768 channels, 840x1297: a width that is not a multiple of 16) forward timing, contiguous output
vs rows padded to 32 pixels (sgs_hip.raster.OUTPUT_PITCH_ALIGN = 32)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-gaussians_amd"))
import torch
from sgs_hip import raster
from sgs_hip.synthetic import CONFIGS, make_config
DEV = "cuda:0"; E = torch.Tensor([])
P, C, W, H, fx = CONFIGS["cfg4"]
scene, cam = make_config("cfg4", features=False)
g = torch.Generator(device=DEV).manual_seed(4)
feats = torch.randn(P, C, device=DEV, generator=g); feats /= feats.norm(dim=1, keepdim=True)
s, c = scene._replace(features=torch.empty(0, C)).to(DEV), cam.to(DEV)
bg = torch.zeros(C, device=DEV)
pool = raster.ScratchPool()
if len(sys.argv) > 1:
    raster.set_binning_mode(int(sys.argv[1]))   # 3 = the library radix sort as the depth presort
def fwd():
    return raster.rasterize_forward(bg, s.means3D, feats, s.opacities, s.scales, s.rotations, 1.0, E, c.world_view_transform,
                                    c.full_proj_transform, c.tanfovx, c.tanfovy, H, W, E, 0, c.camera_center, False, False, C, False, pool=pool)
for align in (0, 32):
    raster.OUTPUT_PITCH_ALIGN = align
    for _ in range(3): fwd()
    torch.cuda.synchronize(); raster.get_stage_ms(); raster.set_stage_timing(2)
    t0 = time.perf_counter()
    for _ in range(10): out = fwd()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
    raster.set_stage_timing(0); ms = raster.get_stage_ms()
    print(f"pitch align {align:2d}: frame {t * 1e3:.3f} ms  stages {[round(x, 3) for x in ms]}  out bytes {C * H * W * 4 / 1e9:.2f} GB  stride {out[1].stride()}", flush=True)

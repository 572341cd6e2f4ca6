"""cfg5 at its stated size on ONE GPU (50 M Gaussians x 256 channels, 968x1296; 51 GB of features): forward frame time and stage times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-gaussians_amd"))
import torch
from sgs_hip import raster
from sgs_hip.synthetic import CONFIGS, make_scene
from sgs_hip.camera import pinhole
DEV = "cuda:0"; E = torch.Tensor([])
P, C, W, H, fx = CONFIGS["cfg5"]
scene = make_scene(P, C, W, H, fx, seed=5, features=False)
s, c = scene.to(DEV), pinhole(W, H, fx).to(DEV)
g = torch.Generator(device=DEV).manual_seed(55)
feats = torch.empty(P, C, device=DEV)
for i in range(0, P, 1 << 21):
    f = torch.randn(min(P, i + (1 << 21)) - i, C, device=DEV, generator=g)
    feats[i:i + f.shape[0]] = f / f.norm(dim=1, keepdim=True)
del f
bg = torch.zeros(C, device=DEV)
pool = raster.ScratchPool()
def fwd():
    return raster.rasterize_forward(bg, s.means3D, feats, s.opacities, s.scales, s.rotations, 1.0, E, c.world_view_transform,
                                    c.full_proj_transform, c.tanfovx, c.tanfovy, H, W, E, 0, c.camera_center, False, False, C, False, pool=pool)
for _ in range(3): out = fwd()
torch.cuda.synchronize(); raster.get_stage_ms(); raster.set_stage_timing(2)
t0 = time.perf_counter()
for _ in range(8): out = fwd()
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 8
raster.set_stage_timing(0); ms = raster.get_stage_ms()
print(f"cfg5 on one GPU: num_rendered {out[0]}, frame {t * 1e3:.3f} ms, stages (preprocess, sort+counts, -, span partitions, -, weights, sweep) {[round(x, 3) for x in ms]}", flush=True)

import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "semantic-gaussians_amd")); sys.path.insert(0, ROOT)
import torch
from sgs_hip import raster, _lib
from sgs_hip.synthetic import CONFIGS, make_scene
from bench import view_camera
DEV = "cuda:0"; E = torch.Tensor([])
P, C, W, H, fx = CONFIGS["cfg4"]
scene = make_scene(P, C, W, H, fx, seed=4, features=False)
s = scene.to(DEV)
g = torch.Generator(device=DEV).manual_seed(44)
feats = torch.randn(P, C, device=DEV, generator=g); feats /= feats.norm(dim=1, keepdim=True)
bg = torch.zeros(C, device=DEV)
views = [view_camera(i, W, H, fx).to(DEV) for i in range(8)]
pool = raster.ScratchPool()
raster.OUTPUT_PITCH_ALIGN = 32
def fwd(c):
    return raster.rasterize_forward(bg, s.means3D, feats, s.opacities, s.scales, s.rotations, 1.0, E, c.world_view_transform,
                                    c.full_proj_transform, c.tanfovx, c.tanfovy, H, W, E, 0, c.camera_center, False, False, C, False, pool=pool)
for p in range(3):
    for i, c in enumerate(views):
        torch.cuda.synchronize(); raster.get_stage_ms(); raster.set_stage_timing(2)
        t0 = time.perf_counter(); o = fwd(c); torch.cuda.synchronize(); t = time.perf_counter() - t0
        raster.set_stage_timing(0); ms = raster.get_stage_ms()
        print(f"pass {p} view {i}: {t*1e3:.2f} ms n={o[0]} stages {[round(x,2) for x in ms]} ovf={raster.stream_stat(_lib.STAT_FWD_OVERFLOWS)} slots={raster.stream_stat(_lib.STAT_ARENA_SLOTS)}", flush=True)

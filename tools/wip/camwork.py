import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/semantic-gaussians_amd")
import numpy as np, torch, bench
from sgs_hip import raster
from sgs_hip.synthetic import CONFIGS, make_scene
P, C, W, H, fx = CONFIGS["cfg3"]
s = make_scene(P, 128, W, H, fx, seed=0).to("cuda:0")
E = torch.Tensor([])
for n in list(range(0, 16)) + [40, 64, 100, 127]:
    c = bench.view_camera(n, W, H, fx).to("cuda:0")
    out = raster.rasterize_forward(s.bg, s.means3D, s.features, s.opacities, s.scales, s.rotations, 1.0, E, c.world_view_transform,
                                   c.full_proj_transform, c.tanfovx, c.tanfovy, H, W, E, 0, c.camera_center, False, False, 128, False)
    iv = raster.image_views(out[5], W, H)
    print(n, "num_rendered", out[0], "visible", int((out[2] > 0).sum()), "sum n_contrib", int(iv["n_contrib"].sum()), "covered px", float((iv["final_T"] < 0.999).float().mean()))

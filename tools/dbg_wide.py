import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-gaussians_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from helpers import small_scene, oracle_forward
from oracle import oracle as orc
from sgs_hip import raster
DEV="cuda:0"; E=torch.Tensor([])
W_, H_ = int(sys.argv[1]), int(sys.argv[2])
scene, cam = small_scene(P=2500, C=128, W=W_, H=H_, fx=90.0)
fw = oracle_forward(orc, scene, cam)
s, c = scene.to(DEV), cam.to(DEV)
raster.set_blend_variant(int(sys.argv[3]) if len(sys.argv) > 3 else 15)
out = raster.rasterize_forward(s.bg, s.means3D, s.features, s.opacities, s.scales, s.rotations, 1.0, E, c.world_view_transform, c.full_proj_transform, c.tanfovx, c.tanfovy, H_, W_, E, 0, c.camera_center, False, False, 128, False)[1].cpu().numpy()
bad = np.abs(out - fw["out"]) > 1e-4
print("bad elems", bad.sum(), "of", bad.size)
cs = np.nonzero(bad.any((1,2)))[0]; ys = np.nonzero(bad.any((0,2)))[0]; xs = np.nonzero(bad.any((0,1)))[0]
print("channels", cs[:40]); print("rows", ys[:40]); print("cols", xs[:64])
# find for a bad element where its value exists in oracle
c0,y0,x0 = np.argwhere(bad)[0]
v = out[c0,y0,x0]
loc = np.argwhere(fw["out"] == v)
print("first bad", c0,y0,x0, "value found in oracle at", loc[:5])
for (c1,y1,x1) in np.argwhere(bad)[:12]:
    loc = np.argwhere(fw["out"] == out[c1,y1,x1]); print((c1,y1,x1), "<-", loc[:2].tolist())

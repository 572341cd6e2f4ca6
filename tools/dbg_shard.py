import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/semantic-gaussians_amd")
import torch
from sgs_hip import raster, dist as sdist
from sgs_hip.synthetic import CONFIGS, make_scene
from sgs_hip.camera import pinhole
DEV="cuda:0"; E=torch.Tensor([])
_, C, W, H, fx = CONFIGS["cfg5"]
C = int(sys.argv[1]) if len(sys.argv) > 1 else C
P = 2_000_000
scene = make_scene(P, C, W, H, fx, seed=5); cam = pinhole(W, H, fx)
s, c = scene.to(DEV), cam.to(DEV)
bg = torch.linspace(0.0, 1.0, C, device=DEV)
for exact in (False, True):
    raster.set_blend_exact(exact)
    n, whole, radii, geom, binn, img, _ = raster.rasterize_forward(bg, s.means3D, s.features, s.opacities, s.scales, s.rotations, 1.0, E, c.world_view_transform, c.full_proj_transform, c.tanfovx, c.tanfovy, H, W, E, 0, c.camera_center, False, False, C, False)
    Tw = raster.image_views(img, W, H)["final_T"].clone()
    depth = s.means3D[:, 2]; cut = float(depth.median())
    parts = []
    for mask in (depth < cut, depth >= cut):
        A, T, r = raster.render_partial(s.means3D[mask], s.features[mask], s.opacities[mask], s.scales[mask], s.rotations[mask], c.world_view_transform, c.full_proj_transform, c.tanfovx, c.tanfovy, H, W, c.camera_center)
        parts.append((A, T))
    comp, tt = sdist.composite_over(parts, bg)
    err = (comp - whole).abs()
    e_pix = err.amax(0)
    idx = int(e_pix.argmax()); y, x = idx // W, idx % W
    print("exact", exact, "max err", float(err.max()), "scale", float(whole.abs().max()), "at", y, x, "Twhole", float(Tw[y, x]), "T1", float(parts[0][1][y, x]), "T2", float(parts[1][1][y, x]), "Tcomp", float(tt[y, x]))
    print("   T diff max", float((tt - Tw).abs().max()), " frac px err>1e-4:", float((e_pix > 1e-4).float().mean()))
    # without bg
    comp0, _ = sdist.composite_over(parts, None)
    print("   bg term at that px:", float(bg.max() * tt[y, x]), "bg*Twhole", float(Tw[y, x]))
raster.set_blend_exact(False)

"""Bring-up of the fused forward blend (variants 32 / 33) against the oracle + timing at cfg3."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-gaussians_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
from helpers import small_scene, oracle_forward
from oracle import oracle as orc
from sgs_hip import raster
from sgs_hip.synthetic import make_config

DEV = "cuda:0"
FV = (35, 34)   # (exact, split-bf16) variants under test
E = torch.Tensor([])


def fwd(scene, cam, variant, pool=None):
    raster.set_blend_variant(variant)
    s, c = scene, cam
    out = raster.rasterize_forward(s.bg, s.means3D, s.features, s.opacities, s.scales, s.rotations, 1.0, E,
                                   c.world_view_transform, c.full_proj_transform, c.tanfovx, c.tanfovy,
                                   c.image_height, c.image_width, E, 0, c.camera_center, False, False,
                                   s.features.shape[1], False, pool=pool)
    raster.set_blend_variant(0)
    return out


def small(C, W, H, P=2500, fx=90.0, seed=0, bg=None):
    scene, cam = small_scene(P=P, C=C, W=W, H=H, fx=fx, seed=seed)
    if bg is not None:
        scene = scene._replace(bg=torch.tensor(bg, dtype=torch.float32))
    fw = oracle_forward(orc, scene, cam)
    s, c = scene.to(DEV), cam.to(DEV)
    res = {}
    for v in FV:
        n, color, radii, geom, binn, img, _ = fwd(s, c, v)
        torch.cuda.synchronize()
        out = color.cpu().numpy()
        iv = raster.image_views(img, W, H)
        okn = np.array_equal(iv["n_contrib"].cpu().numpy().view(np.uint32), fw["n_contrib"])
        okT = np.array_equal(iv["final_T"].cpu().numpy().view(np.uint32), fw["final_T"].view(np.uint32))
        err = np.abs(out - fw["out"]).max() / (np.abs(fw["out"]).max() + 1e-30)
        bad = int((out != fw["out"]).sum())
        res[v] = (err, bad, okn, okT)
        if v in (33, 35) and bad:
            ys, xs = np.nonzero((out != fw["out"]).any(0))
            cs = np.nonzero((out != fw["out"]).any((1, 2)))[0]
            print("   mismatch px rows", sorted(set(ys.tolist()))[:20], "cols", sorted(set(xs.tolist()))[:20], "ch", cs[:8], "nan", int(np.isnan(out).sum()))
    print(f"C={C} {W}x{H} P={P}: exact rel={res[FV[0]][0]:.2e} diff_elems={res[FV[0]][1]} n_contrib={res[FV[0]][2]} T={res[FV[0]][3]} | "
          f"bf16 rel={res[FV[1]][0]:.2e} n_contrib={res[FV[1]][2]} T={res[FV[1]][3]}", flush=True)
    return res


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "small"):
        small(128, 64, 48, P=800, fx=60.0)
        small(128, 208, 160)
        small(256, 200, 120, bg=np.linspace(-1, 1, 256))
        small(512, 192, 100)
        small(128, 400, 70, P=6000)
        small(384, 100, 100)
    if what in ("all", "cfg3"):
        scene, cam = make_config("cfg3")
        s, c = scene.to(DEV), cam.to(DEV)
        pool = raster.ScratchPool()
        for v in [int(x, 0) for x in (sys.argv[2:] or ["0", "0x408", "15"])]:
            for _ in range(3):
                fwd(s, c, v, pool)
            torch.cuda.synchronize()
            raster.get_stage_ms()
            raster.set_stage_timing(2)
            t0 = time.perf_counter()
            for _ in range(10):
                out = fwd(s, c, v, pool)
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / 10
            raster.set_stage_timing(0)
            ms = raster.get_stage_ms()
            print(f"variant {v:#x}: frame {t * 1e3:.3f} ms  stages {[round(x, 3) for x in ms]}  blend {ms[5] + ms[6]:.3f} ms", flush=True)
            if v == 0:
                ref = out[1].clone()
            else:
                d = (out[1] - ref).abs().max().item() / ref.abs().max().item()
                print(f"   vs default: max rel diff {d:.2e}")

"""Reproducer for DESIGN.md 5.4's finding: with the gfx950 double-rate v_mfma_f32_32x32x16_bf16 in the sweep kernel,
forwards running BESIDE it on other HIP streams sporadically come out with a few wrong radii (one aligned 256-byte
beat of a global_load_dwordx4 in the victim's preprocess), about one forward in 1000; with the same products issued
as pairs of v_mfma_f32_32x32x8_bf16_1k (the shipped kernel) none.

Each round renders 6 views of a small C = 128 scene with 4 in flight on 4 streams and compares every view's
num_rendered and radii with the serial result.  Blend variant 0x808 selects the x16 build of the otherwise identical
sweep kernel (blend_fwd_split.hip, DBG bit 8).   usage: repro_x16_neighbour_corruption.py [rounds=4000]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "semantic-gaussians_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from helpers import small_scene
from sgs_hip import raster, dist as sdist
from sgs_hip.camera import pinhole

DEV = "cuda:0"
scene, _ = small_scene(P=5000, C=128, W=208, H=128, fx=170.0, seed=5)
s = scene.to(DEV)
cams = [pinhole(208, 128, fx).to(DEV) for fx in (150.0, 160.0, 170.0, 180.0, 190.0, 200.0)]
e = torch.Tensor([])
pool = raster.ScratchPool()


def render(c, slot):
    out = raster.rasterize_forward(s.bg, s.means3D, s.features, s.opacities, s.scales, s.rotations, 1.0, e,
                                   c.world_view_transform, c.full_proj_transform, c.tanfovx, c.tanfovy,
                                   128, 208, e, 0, c.camera_center, False, False, 128, False, pool=pool)
    return out[0], out[2].clone(), out[3].clone()


def run(name, variant, rounds):
    raster.set_blend_variant(variant)
    serial = [render(c, 0) for c in cams]
    torch.cuda.synchronize()
    bad = 0
    for _ in range(rounds):
        piped = sdist.render_views_pipelined(render, cams, in_flight=4)
        for vi, (a, b) in enumerate(zip(serial, piped)):
            wrong = a[0] != b[0] or not torch.equal(a[1], b[1])
            bad += int(wrong)
            if wrong and bad <= 6:   # what does the damage look like?  (indices, values as int32 and as float bits)
                d = torch.nonzero(a[1] != b[1]).flatten()
                got = b[1][d].cpu()
                ga, gb = raster.geometry_views(a[2], 5000), raster.geometry_views(b[2], 5000)
                for k in ("depths", "means2D", "cov3D", "conic_opacity", "tiles_touched"):
                    va, vb = ga[k].reshape(5000, -1), gb[k].reshape(5000, -1)
                    dd = torch.nonzero((va != vb).any(dim=1)).flatten()
                    if dd.numel():
                        i0 = int(dd[0])
                        print(f"      {k}: {dd.numel()} rows differ, first {i0}: want {va[i0].tolist()} got {vb[i0].tolist()}", flush=True)
                print(f"  [{name}] view {vi}: num_rendered {a[0]} vs {b[0]}; {d.numel()} radii differ at {d[:24].tolist()}"
                      f" want {a[1][d][:8].tolist()} got {got[:8].tolist()} as f32 {got[:8].view(torch.float32).tolist()}", flush=True)
    raster.set_blend_variant(0)
    print(f"{name}: {bad} corrupted forwards of {rounds * 6}", flush=True)
    return bad


if __name__ == "__main__":
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    if len(sys.argv) > 2:   # only the named variants, e.g. "0x808 0x6C"
        for a in sys.argv[2:]:
            run(a, int(a, 0), R)
        sys.exit(0)
    run("x8 pairs (shipped)", 0, R)
    run("x16 (v_mfma_f32_32x32x16_bf16)", 0x808, R)
    run("sweep2, six products on x16 (0x6C)", 0x6C, R)
    run("sweep2, six products on x8 pairs (0x6E)", 0x6E, R)
    run("x8 pairs again", 0, R)

"""Stress of concurrent forwards: N rounds of 6 views with 4 in flight on 4 HIP streams, each compared (num_rendered,
radii) with the serial result.  This is the run that exposed the cross-kernel corruption associated with
v_mfma_f32_32x32x16_bf16 in the sweep kernel.  usage: stress_pipelined.py [rounds]"""
import sys
sys.path.insert(0, "/root/repo/semantic-gaussians_amd"); sys.path.insert(0, "/root/repo/tests")
import torch
from helpers import small_scene
from sgs_hip import raster, dist as sdist
from sgs_hip.camera import pinhole
DEV = "cuda:0"
scene, cam0 = small_scene(P=5000, C=128, W=208, H=128, fx=170.0, seed=5)
s = scene.to(DEV)
cams = [pinhole(208, 128, fx).to(DEV) for fx in (150.0, 160.0, 170.0, 180.0, 190.0, 200.0)]
e = torch.Tensor([])
pool = raster.ScratchPool()
copies = {}
def render(c, slot, percopy=False, sync=False, deferred=False):
    sc = s
    if percopy:
        k = torch.cuda.current_stream().cuda_stream
        if k not in copies:
            copies[k] = s._replace(scales=s.scales.clone(), rotations=s.rotations.clone(), means3D=s.means3D.clone(), opacities=s.opacities.clone())
        sc = copies[k]
    fn = raster.rasterize_forward_deferred if deferred else raster.rasterize_forward
    out = fn(sc.bg, sc.means3D, sc.features, sc.opacities, sc.scales, sc.rotations, 1.0, e,
             c.world_view_transform, c.full_proj_transform, c.tanfovx, c.tanfovy,
             128, 208, e, 0, c.camera_center, False, False, 128, False, pool=pool)
    if deferred:   # (resolved by render_views_pipelined; radii are read after its final synchronize)
        return out
    r = out[0], out[2].clone()
    if sync: torch.cuda.synchronize()
    return r
def run(name, reps=150, **kw):
    serial = [render(c, 0, **kw) for c in cams]
    torch.cuda.synchronize()
    bad = 0
    for rep in range(reps):
        piped = sdist.render_views_pipelined(lambda c, sl: render(c, sl, **kw), cams, in_flight=2)
        for a, c in zip(serial, piped):
            bad += int(a[0] != c[0] or not torch.equal(a[1], c[1]))
    print(name, "bad", bad, "of", reps * 6, flush=True)
import sys
NF = 4
def run(name, reps=600, **kw):
    serial = [render(c, 0, **{**kw, "deferred": False}) for c in cams]
    torch.cuda.synchronize()
    bad = 0
    for rep in range(reps):
        piped = sdist.render_views_pipelined(lambda c, sl: render(c, sl, **kw), cams, in_flight=NF)
        for a, c in zip(serial, piped):
            bad += int(a[0] != c[0] or not torch.equal(a[1], c[2] if len(c) > 2 else c[1]))
    print(name, "bad", bad, "of", reps * 6, flush=True)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
run("default", R)
run("default, deferred counts", R, deferred=True)
raster.set_blend_variant(15); run("exact sweep", R); raster.set_blend_variant(0)

import sys, time, os
sys.path.insert(0, "semantic-gaussians_amd")
import torch
from sgs_hip import raster
from sgs_hip.synthetic import make_config
scene, cam = make_config("cfg3", C=128)
dev = "cuda:0"
s, c = scene.to(dev), cam.to(dev)
e = torch.Tensor([])
orig = raster._Buffers.callback
def cb(self, key):
    inner = orig(self, key)
    return inner
import ctypes
# monkeypatch torch.empty timing inside callback
real_empty = torch.empty
def timed_empty(*a, **k):
    t = time.perf_counter(); r = real_empty(*a, **k); dt = time.perf_counter() - t
    if dt > 1e-3: print("  slow empty", a[0] if a else None, f"{dt*1e3:.2f} ms", flush=True)
    return r
torch.empty = timed_empty
for i in range(12):
    torch.cuda.synchronize(); t = time.perf_counter()
    out = raster.rasterize_forward(s.bg, s.means3D, s.features, s.opacities, s.scales, s.rotations, 1.0, e, c.world_view_transform, c.full_proj_transform, c.tanfovx, c.tanfovy, 968, 1296, e, 0, c.camera_center, False, False, 128, False)
    torch.cuda.synchronize(); print(i, f"{(time.perf_counter()-t)*1e3:.2f} ms", "reserved GB", torch.cuda.memory_reserved()/1e9, "binning bytes", out[4].numel(), flush=True)

"""Where a wave of the ping-pong sweep (blend_accum_sweep3_kernel) spends its cycles at cfg3: the kernel's DBG & 4 build
stamps s_memtime at every phase boundary of a step and sums the differences per wave (blend_sweep2.hip S3_STAMP).
    python tools/sweep_phases.py [variant=0x466]
Prints, per half of the workgroup (waves 0-3 = parity 0, 4-7 = parity 1), the mean shader cycles per STEP in each phase."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-gaussians_amd"))
import numpy as np
import torch
from sgs_hip import raster, _lib
from sgs_hip.synthetic import CONFIGS, make_scene
from sgs_hip.camera import pinhole

dev = "cuda:0"
variant = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0x466
P, C, W, H, fx = CONFIGS["cfg3"]
s = make_scene(P, C, W, H, fx, seed=0).to(dev)
c = pinhole(W, H, fx).to(dev)
empty = torch.empty(0, device=dev)
raster.set_blend_variant(variant)


def fwd():
    return raster.rasterize_forward(s.bg, s.means3D, s.features, s.opacities, s.scales, s.rotations, 1.0, empty,
                                    c.world_view_transform, c.full_proj_transform, c.tanfovx, c.tanfovy, H, W, empty,
                                    0, c.camera_center, False, False, C, False)


for _ in range(3):
    fwd()
torch.cuda.synchronize()
NWG = 4096
BASE = 4 * (NWG + 8192)
tr = torch.zeros(BASE + 12 * 8 * NWG, dtype=torch.int64, device=dev)
lib = _lib.load()
lib.sgs_debug_set_sweep_trace(tr.data_ptr())
fwd()
torch.cuda.synchronize()
lib.sgs_debug_set_sweep_trace(None)
raster.set_blend_variant(0)
t = tr.cpu().numpy()
wg = t[:4 * NWG].reshape(NWG, 4)
ph = t[BASE:].reshape(NWG * 8, 12)
ph = ph[ph[:, 10] != 0]
names = ["table words", "DMA issue (5 pieces)", "deferred stores", "operand wait + split", "arrival check (half 1)", "barrier 1",
         "48 MFMAs", "arrival check (half 0)", "barrier 2", "between steps (pair stores ...)"]
used = wg[wg[:, 1] != 0]
span = (used[:, 1].max() - used[:, 0].min()) / 100.0
print(f"variant {variant:#x}: {len(used)} workgroups, kernel span {span:.1f} us (with the stamps' own cost), {len(ph)} wave records")
for half in (0, 1):
    sel = ph[(ph[:, 11] >> 2) == half]
    steps = sel[:, 10].sum()
    tot = sel[:, :10].sum()
    print(f"half {half} (waves {4 * half}-{4 * half + 3}): {tot / steps:8.0f} cycles per step")
    for k in range(10):
        print(f"    {names[k]:34s} {sel[:, k].sum() / steps:8.0f}  ({100.0 * sel[:, k].sum() / tot:4.1f} %)")

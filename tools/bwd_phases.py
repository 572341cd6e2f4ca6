"""Where a wave of the fused backward kernel (bwd_fused_kernel, blend_bwd_mfma.hip) spends its cycles at cfg3: the DBG & 16 build
stamps s_memtime at the phase boundaries of an iteration and sums the differences per wave.
    SGS_BWD_DBG=16 python tools/bwd_phases.py [backward_mode=0]
Prints, per half of the workgroup (waves 0-3 = lower K half, 4-7 = upper), the mean shader cycles per iteration in each phase."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-gaussians_amd"))
import numpy as np
import torch
from sgs_hip import raster, _lib
from sgs_hip.synthetic import CONFIGS, make_scene
from sgs_hip.camera import pinhole

assert os.environ.get("SGS_BWD_DBG") in ("16", "48"), "run with SGS_BWD_DBG=16 (one workgroup per tile) or 48 (persistent workgroups)"
dev = "cuda:0"
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
P, C, W, H, fx = CONFIGS["cfg3"]
s = make_scene(P, C, W, H, fx, seed=0).to(dev)
c = pinhole(W, H, fx).to(dev)
empty = torch.empty(0, device=dev)
dL = torch.randn(C, H, W, device=dev)
raster.set_backward_mode(mode)
lib = _lib.load()
NWG = 8192
tr = torch.zeros(24 * 8 * NWG, dtype=torch.int64, device=dev)


def fwd_bwd():
    n, col, rad, g_, b_, i_, _ = raster.rasterize_forward(
        s.bg, s.means3D, s.features, s.opacities, s.scales, s.rotations, 1.0, empty, c.world_view_transform,
        c.full_proj_transform, c.tanfovx, c.tanfovy, H, W, empty, 0, c.camera_center, False, False, C, False)
    return raster.rasterize_backward(s.bg, s.means3D, rad, s.features, s.scales, s.rotations, 1.0, empty,
                                     c.world_view_transform, c.full_proj_transform, c.tanfovx, c.tanfovy, dL,
                                     empty, 0, c.camera_center, g_, n, b_, i_, False)


for _ in range(2):
    fwd_bwd()
torch.cuda.synchronize()
n, col, rad, g_, b_, i_, _ = raster.rasterize_forward(
    s.bg, s.means3D, s.features, s.opacities, s.scales, s.rotations, 1.0, empty, c.world_view_transform,
    c.full_proj_transform, c.tanfovx, c.tanfovy, H, W, empty, 0, c.camera_center, False, False, C, False)
torch.cuda.synchronize()
lib.sgs_debug_set_sweep_trace(tr.data_ptr())
raster.rasterize_backward(s.bg, s.means3D, rad, s.features, s.scales, s.rotations, 1.0, empty, c.world_view_transform,
                          c.full_proj_transform, c.tanfovx, c.tanfovy, dL, empty, 0, c.camera_center, g_, n, b_, i_, False)
torch.cuda.synchronize()
lib.sgs_debug_set_sweep_trace(None)
raster.set_backward_mode(0)
ph = tr.cpu().numpy().reshape(NWG * 8, 24)
ph = ph[ph[:, 20] != 0]
names = ["request feature pieces", "W g^T (upper half: first)", "take slab: wait, split, transposing stores", "finish slab s-2 (lower: exchange + atomics)",
         "request gradient", "D products", "W g^T (lower half: last)", "stage features (wait, split, store)", "barrier", "between iterations / chunks (rest)",
         "chunk: first round trip (count, ids, weights, slab 0)", "chunk: barrier, ids to LDS, row pointers", "chunk: weights split",
         "chunk: second round trip (features) + staging", "chunk: barrier", "tail: finish slab n-2", "tail: W g^T of the last slab", "tail: barrier",
         "tail: finish the last slab", "tail: D rows out (drained)"]
tot_entries = (ph[:, 21] >> 8)[::8].sum()
print(f"backward mode {mode}: {len(ph) // 8} workgroups, {ph[:, 20][::8].sum()} iterations, {tot_entries} work-list entries")
for half in (0, 1):
    sel = ph[((ph[:, 21] & 255) >> 2) == half]
    it = sel[:, 20].sum()
    tot = sel[:, :20].sum()
    print(f"half {half} (waves {4 * half}-{4 * half + 3}): {tot / it:8.0f} cycles per iteration")
    for k in range(20):
        print(f"    {names[k]:56s} {sel[:, k].sum() / it:8.0f}  ({100.0 * sel[:, k].sum() / tot:4.1f} %)")

"""Per-workgroup timeline of blend_accum_sweep_kernel at cfg3 (debug hook sgs_debug_set_sweep_trace):
how full the 512 workgroup slots are over the kernel's life, the tail, duration vs work.
    python tools/sweep_trace.py [variant]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-gaussians_amd"))
import numpy as np
import torch
from sgs_hip import raster, _lib
from sgs_hip.synthetic import CONFIGS, make_scene
from sgs_hip.camera import pinhole

dev = "cuda:0"
variant = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0
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
tr = torch.zeros(NWG + 8192, 4, dtype=torch.int64, device=dev)   # sweep workgroups | weights workgroups
lib = _lib.load()
lib.sgs_debug_set_sweep_trace(tr.data_ptr())
fwd()
torch.cuda.synchronize()
lib.sgs_debug_set_sweep_trace(None)
t_all = tr.cpu().numpy()
which = sys.argv[2] if len(sys.argv) > 2 else "sweep"
t = t_all[:NWG] if which == "sweep" else t_all[NWG:]
SLOTS = 512 if which == "sweep" else 1536
print(f"== {which} kernel")
t = t[t[:, 1] != 0]
b, e = t[:, 0].astype(np.float64), t[:, 1].astype(np.float64)
t0 = b.min()
b, e = (b - t0) / 100.0, (e - t0) / 100.0   # us
span = e.max()
dur = e - b
J = t[:, 3] & 0xFFFFFFFF
nt = (t[:, 3] >> 32) & 0xFF
late = t[:, 3] >> 40   # (blend_sweep2: steps of wave 0 that found their bundle still in flight)
hw = t[:, 2] & 0xFFFFFFFF
xcc = (t[:, 2] >> 32) & 0xF
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7
sh = (hw >> 12) & 1
print(f"workgroups {len(t)}, kernel span {span:.1f} us, sum of durations {dur.sum():.0f} us = {dur.sum() / span:.1f} slots busy on average")
if which == "sweep":
    print(f"steps that began before their bundle had landed (wave 0 of each workgroup): {int(late.sum())} of {int(J.sum())} = {late.sum() / max(1, J.sum()):.3f}")
print(f"duration us: min {dur.min():.1f} median {np.median(dur):.1f} max {dur.max():.1f}; batches/WG median {np.median(J):.0f} max {J.max()}")
k = np.polyfit(J, dur, 1)
print(f"duration ~ {k[0]:.3f} us/batch * J + {k[1]:.1f} us;  residual std {np.std(dur - np.polyval(k, J)):.1f} us")
# occupancy over time
grid = np.linspace(0, span, 41)
act = [(np.sum((b <= x) & (e > x))) for x in grid]
print("active WGs at 2.5 % steps of the span:", " ".join(str(a) for a in act))
# start-time rounds
print(f"WGs started in the first 10 us: {np.sum(b < 10)}, after half of the span: {np.sum(b > span / 2)}")
places = {}
for i in range(len(t)):
    places.setdefault((int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i])), []).append((b[i], e[i]))
busy = np.array([sum(y - x for x, y in v) for v in places.values()])
print(f"distinct (xcc, se, sh, cu): {len(places)}; busy us per place: min {busy.min():.0f} median {np.median(busy):.0f} max {busy.max():.0f}"
      f"; WGs per place min {min(len(v) for v in places.values())} max {max(len(v) for v in places.values())}")
last = np.array([max(y for _, y in v) for v in places.values()])
print(f"last end per place us: min {last.min():.0f} median {np.median(last):.0f} max {last.max():.0f}")
np.save(os.path.join(ROOT, "gpurun_out", "sweep_trace.npy"), t) if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None

"""How many (group of 8 / 16 active entries) x (32-pixel MFMA block) weight blocks of the forward's work list are all zero?
(CPU, sampled tiles of cfg3.)  A zero block needs no hand-over bytes and no MFMA work."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "semantic-gaussians_amd")); sys.path.insert(0, ROOT)
import torch
from sgs_hip.synthetic import CONFIGS, make_scene
from sgs_hip.camera import pinhole
from oracle import torch_splat as ts
torch.set_num_threads(32)
P, C, W, H, fx = CONFIGS["cfg3"]
scene = make_scene(P, C, W, H, fx, seed=0, features=False)
cam = pinhole(W, H, fx)
pre = ts.preprocess(scene.means3D, scene.scales, scene.rotations, scene.opacities, cam.world_view_transform, cam.full_proj_transform,
                    W, H, cam.tanfovx, cam.tanfovy)
binn = ts.binning(pre, W, H)
gx, gy = (W + 15) // 16, (H + 15) // 16
pix, conic, opac, plist, ranges = pre["pix"], pre["conic"], pre["opacity"], binn["point_list"], binn["ranges"]
yy, xx = torch.meshgrid(torch.arange(16, dtype=torch.float32), torch.arange(16, dtype=torch.float32), indexing="ij")
yy, xx = yy.reshape(-1, 1), xx.reshape(-1, 1)
# block of pixel (y, x) of a tile: parity g = y & 1, pb = y >> 2  -> rows 4 pb + g and 4 pb + g + 2
blk = ((torch.arange(16) & 1) * 4 + (torch.arange(16) >> 2)).repeat_interleave(16)   # (256,) in 0..7
tot = {8: [0, 0], 16: [0, 0]}
ent_act = ent_all = 0
for t in range(7, gx * gy, 23):
    r0, r1 = int(ranges[t, 0]), int(ranges[t, 1])
    if r1 <= r0:
        continue
    ids = plist[r0:r1]
    px, py = xx + (t % gx) * 16, yy + (t // gx) * 16
    dx, dy = pix[ids, 0][None, :] - px, pix[ids, 1][None, :] - py
    k = conic[ids]
    power = -0.5 * (k[None, :, 0] * dx * dx + k[None, :, 2] * dy * dy) - k[None, :, 1] * dx * dy
    alpha = torch.clamp(opac[ids][None, :] * torch.exp(power), max=0.99)
    ok = (power <= 0) & (alpha >= 1.0 / 255.0) & ((px < W) & (py < H))
    a = torch.where(ok, alpha, torch.zeros_like(alpha))
    Tex = torch.cumprod(torch.cat([torch.ones(256, 1), (1 - a)[:, :-1]], 1), 1)
    stop = ok & (Tex * (1 - a) < 1e-4)
    dead = torch.cumsum(stop.int(), 1) > 0
    a = torch.where(dead, torch.zeros_like(a), a)
    Tex = torch.cumprod(torch.cat([torch.ones(256, 1), (1 - a)[:, :-1]], 1), 1)
    w = a * Tex                                             # (256, n)
    # the kernel walks the list only until every pixel is done
    alive = (~dead).any(0)
    nwalk = int(alive.sum())
    act = (w > 0).any(0)
    ent_all += nwalk; ent_act += int(act.sum())
    wa = w[:, act]
    Tfin = (Tex[:, -1] * (1 - a[:, -1])).reshape(256, 1)
    wa = torch.cat([wa, Tfin], 1)                           # closing pseudo entry
    nz = torch.zeros(8, wa.shape[1], dtype=torch.bool)
    for b in range(8):
        nz[b] = (wa[blk == b] != 0).any(0)
    for gsz in (8, 16):
        n = wa.shape[1]
        pad = (-n) % 16                                     # batches are padded to 16 entries
        z = torch.cat([nz, torch.zeros(8, pad, dtype=torch.bool)], 1).reshape(8, -1, gsz).any(2)   # (8, groups)
        tot[gsz][0] += int((~z).sum()); tot[gsz][1] += z.numel()
print(f"sampled tiles: active entries {ent_act} of {ent_all} walked")
for gsz in (8, 16):
    print(f"groups of {gsz:2d} entries x 32-pixel blocks: {tot[gsz][0]} of {tot[gsz][1]} all zero = {tot[gsz][0] / tot[gsz][1]:.3f}")

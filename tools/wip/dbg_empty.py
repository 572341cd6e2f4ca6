import sys, os
sys.path.insert(0, "/root/repo/semantic-gaussians_amd"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import torch, numpy as np
from helpers import small_scene
from test_parity_gpu import _hip_forward
scene, cam = small_scene(P=60, C=128, W=208, H=96, fx=170.0, seed=5)
g = torch.Generator().manual_seed(3)
scene = scene._replace(bg=torch.randn(128, generator=g), scales=scene.scales * 0.3)
ref = _hip_forward(scene, cam, variant=15)[1].cpu().numpy()
for v in (0x6B, 0x6E):
    out = _hip_forward(scene, cam, variant=v)[1].cpu().numpy()
    d = (out != ref) if v == 0x6B else (np.abs(out - ref) > 1e-5)
    print(hex(v), "differing elements", d.sum(), "of", d.size)
    ch, ys, xs = np.nonzero(d)
    if len(ch):
        print(" channels", np.unique(ch)[:20], "... count", len(np.unique(ch)))
        print(" rows", np.unique(ys)); print(" tile cols", np.unique(xs // 16)); print(" x within tile", np.unique(xs % 16)[:16])
        tiles = sorted(set(zip((ys // 16).tolist(), (xs // 16).tolist())))
        print(" tiles", tiles[:40])
        i = 0; print(" sample", ch[i], ys[i], xs[i], out[ch[i], ys[i], xs[i]], ref[ch[i], ys[i], xs[i]], "bg", float(scene.bg[ch[i]]))
from helpers import oracle_forward
from oracle import oracle as orc
orc.lib()
fw = oracle_forward(orc, scene, cam)
out = _hip_forward(scene, cam, variant=15)[1].cpu().numpy()
d = out.view(np.uint32) != fw["out"].view(np.uint32)
print("r2 exact vs oracle: differing", d.sum())
ch, ys, xs = np.nonzero(d)
if len(ch):
    print(" channels", len(np.unique(ch)), " rows", np.unique(ys)[:30], " cols", np.unique(xs)[:30])
    for i in range(0, min(len(ch), 2000), 400):
        c, y, x = ch[i], ys[i], xs[i]
        print("  ", c, y, x, "hip", out[c, y, x], "orc", fw["out"][c, y, x], "bg", float(scene.bg[c]), "T", fw["final_T"][y, x], "n_contrib", fw["n_contrib"][y, x], "diff", out[c,y,x]-fw["out"][c,y,x])
    r = fw["ranges"].reshape(-1, 2); print("empty tiles:", int((r[:,0]==r[:,1]).sum()), "of", len(r))
for v in (1, 0, 15, 0x1F, 0x6F):
    o = _hip_forward(scene, cam, variant=v)[1].cpu().numpy()
    bad = np.abs(o - fw["out"]) > 1e-3
    print("variant", hex(v), "bad elements", int(bad.sum()))
r = fw["ranges"].reshape(-1, 2)
t = 5 * 13 + 3
print("tile", t, "range", r[t], "prev", r[t-1], "next", r[t+1])
pl = fw["point_list"][r[t][0]:r[t][1]]
print("ids", pl, "opac", fw["conic_opacity"][pl][:, 3], "xy", fw["means2D"][pl])

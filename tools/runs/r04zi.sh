#!/bin/bash
# round 4, GPU call ZI: two-pixels-per-lane pre-pass with the super-batch walk (0x4036) against the super-batch kernel (0x36) and the two-pixels-per-lane kernel (0xC036)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04zi; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python - > $O/check.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, "semantic-gaussians_amd"); sys.path.insert(0, "tests")
import torch
from helpers import small_scene
from test_parity_gpu import _hip_forward
from sgs_hip import raster
for (P, C, W, H, fx, seed, sc, op) in ((300, 128, 64, 48, 100.0, 2, 1.0, 1.0), (6000, 256, 400, 160, 300.0, 21, 1.0, 1.0), (3000, 128, 336, 48, 170.0, 4, 1.0, 1.0), (40000, 128, 784, 32, 600.0, 77, 3.0, 0.05), (60000, 128, 96, 64, 90.0, 5, 0.6, 0.08), (20000, 128, 48, 48, 60.0, 9, 2.0, 0.02), (500, 512, 48, 40, 170.0, 3, 1.0, 1.0)):
    scene, cam = small_scene(P=P, C=C, W=W, H=H, fx=fx, seed=seed)
    g = torch.Generator().manual_seed(seed)
    scene = scene._replace(bg=torch.randn(C, generator=g), scales=scene.scales * sc, opacities=scene.opacities * op)
    for _ in range(3): _hip_forward(scene, cam, variant=0x66)
    a = _hip_forward(scene, cam, variant=0x66)
    b = _hip_forward(scene, cam, variant=0x4066)
    ia, ib = raster.image_views(a[5], W, H), raster.image_views(b[5], W, H)
    print(P, C, W, H, "bitwise equal:", bool(torch.equal(a[1], b[1])), bool(torch.equal(ia["n_contrib"], ib["n_contrib"])), bool(torch.equal(ia["final_T"], ib["final_T"])), flush=True)
PY
grep -v amdgpu $O/check.txt
timeout 300 python tools/exp_r03_sweep2.py 0x36 0x4036 0xC036 0x36 0x4036 0xC036 0x36 0x4036 > $O/timing.txt 2>&1; grep frame $O/timing.txt

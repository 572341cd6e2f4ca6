#!/bin/bash
# round 4, GPU call W: the ping-pong sweep without barriers (sweep nibble 4: free-running halves, arrival / consumption counters per ring stage)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04w; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python - > $O/check.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, "semantic-gaussians_amd"); sys.path.insert(0, "tests")
import torch
from helpers import small_scene
from test_parity_gpu import _hip_forward
for (P, C, W, H, fx, seed, sc) in ((6000, 256, 400, 160, 300.0, 21, 1.0), (3000, 128, 336, 48, 170.0, 4, 1.0), (40000, 128, 784, 32, 600.0, 77, 3.0), (500, 512, 48, 40, 170.0, 3, 1.0), (4000, 256, 208, 96, 170.0, 1, 1.0)):
    scene, cam = small_scene(P=P, C=C, W=W, H=H, fx=fx, seed=seed)
    g = torch.Generator().manual_seed(seed)
    scene = scene._replace(bg=torch.randn(C, generator=g), scales=scene.scales * sc, opacities=scene.opacities * (0.05 if sc > 1 else 1.0))
    for segn in (3, 1, 6):
        a = _hip_forward(scene, cam, variant=0x6 | (segn << 4))[1]
        b = _hip_forward(scene, cam, variant=0x4 | (segn << 4))[1]
        print(P, C, W, H, "seg", segn, "bitwise equal to the lock-step form:", bool(torch.equal(a, b)), float((a - b).abs().max()), flush=True)
PY
grep -v amdgpu.ids $O/check.txt | tail -16
timeout 300 python tools/exp_r03_sweep2.py 0x36 0x34 0x36 0x34 0x36 0x34 0x134 0x234 > $O/timing.txt 2>&1; grep frame $O/timing.txt

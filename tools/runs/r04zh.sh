#!/bin/bash
# round 4, GPU call ZH: the two-pixels-per-lane pre-pass (0xC036) with whole-line nt stores in its three-term flush, against the super-batch kernel (0x36)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04zh; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_sweep2_gpu.py -q -m gpu -x -k "superbatch or ping_pong" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.txt
timeout 200 python - > $O/check.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, "semantic-gaussians_amd"); sys.path.insert(0, "tests")
import torch
from helpers import small_scene
from test_parity_gpu import _hip_forward
for (P, C, W, H, fx, seed, sc) in ((6000, 256, 400, 160, 300.0, 21, 1.0), (3000, 128, 336, 48, 170.0, 4, 1.0), (40000, 128, 784, 32, 600.0, 77, 3.0), (500, 512, 48, 40, 170.0, 3, 1.0)):
    scene, cam = small_scene(P=P, C=C, W=W, H=H, fx=fx, seed=seed)
    g = torch.Generator().manual_seed(seed)
    scene = scene._replace(bg=torch.randn(C, generator=g), scales=scene.scales * sc, opacities=scene.opacities * (0.05 if sc > 1 else 1.0))
    for _ in range(2): _hip_forward(scene, cam, variant=0x36)
    a = _hip_forward(scene, cam, variant=0x36)[1]
    b = _hip_forward(scene, cam, variant=0xC036)[1]
    print(P, C, W, H, "two-pixels-per-lane pre-pass bitwise equal:", bool(torch.equal(a, b)), flush=True)
PY
grep -v amdgpu $O/check.txt
timeout 300 python tools/exp_r03_sweep2.py 0x36 0xC036 0x36 0xC036 0x36 0xC036 > $O/timing.txt 2>&1; grep frame $O/timing.txt

#!/bin/bash
# round 4, GPU call K (library built with make X16=1): does SPACING the double-rate MFMAs remove the neighbour corruption?
# ping-pong sweep on x16: dense (0x1....), one s_nop between MFMAs (0x2....), one VALU move between MFMAs (0x3....)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r04k; mkdir -p $O
export TMPDIR=/tmp
python - > $O/check.txt 2>&1 <<'PY'
import sys, os
sys.path.insert(0, "semantic-gaussians_amd"); sys.path.insert(0, "tests")
import torch
from helpers import small_scene
from test_parity_gpu import _hip_forward
scene, cam = small_scene(P=6000, C=256, W=400, H=160, fx=300.0, seed=21)
ref = _hip_forward(scene, cam, variant=0x36)[1]
for v in (0x10036, 0x20036, 0x30036):
    o = _hip_forward(scene, cam, variant=v)[1]
    print(hex(v), "max |diff| / max |ref| =", float((o - ref).abs().max() / ref.abs().max()), "identical to each other:", None)
a = _hip_forward(scene, cam, variant=0x10036)[1]; b = _hip_forward(scene, cam, variant=0x20036)[1]; c = _hip_forward(scene, cam, variant=0x30036)[1]
print("x16 forms bitwise equal:", bool(torch.equal(a, b) and torch.equal(a, c)))
PY
cat $O/check.txt | grep -v amdgpu.ids
timeout 300 python tools/exp_r03_sweep2.py 0x36 0x10036 0x20036 0x30036 0x36 0x10036 0x20036 0x30036 > $O/timing.txt 2>&1; grep frame $O/timing.txt
timeout 600 python tools/repro_x16_neighbour_corruption.py 4000 0x36 0x10036 0x20036 0x30036 0x10036 > $O/repro_short.txt 2>&1; grep "corrupted" $O/repro_short.txt
timeout 900 python tools/repro_x16_neighbour_corruption.py 20000 0x20036 0x30036 0x10036 > $O/repro_long.txt 2>&1; grep "corrupted" $O/repro_long.txt

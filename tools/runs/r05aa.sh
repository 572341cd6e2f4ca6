#!/bin/bash
# round 5, GPU call AA (EXPERIMENTS=1 build): the ping-pong sweep's free-running form (flags instead of barriers) on the x16 MFMA against the lock-step default
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05aa; mkdir -p $O
python - <<'PY'
import sys, os
sys.path.insert(0, "semantic-gaussians_amd"); sys.path.insert(0, "tests")
import torch
from helpers import small_scene
from test_parity_gpu import _hip_forward
for (P, C, W, H, fx, seed) in ((4000, 256, 208, 96, 170.0, 1), (30000, 128, 400, 64, 170.0, 2), (40000, 128, 784, 32, 600.0, 77)):
    scene, cam = small_scene(P=P, C=C, W=W, H=H, fx=fx, seed=seed)
    for _ in range(2): _hip_forward(scene, cam, variant=0)
    a = _hip_forward(scene, cam, variant=0x10066)[1]; b = _hip_forward(scene, cam, variant=0x10064)[1]
    print("free-running x16 == lock-step x16:", bool(torch.equal(a, b)), (P, C, W, H))
PY
for v in 0 65540 0 65540; do python bench.py --variant $v --no-cpu-baseline --no-extras --steps 120 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant', hex($v), 'value', round(d['value'],1), 'single', round(d['single_view']['ms_median'],4), 'roofline', round(d['roofline']['frac'],4), d['roofline']['kernels_ms'], d['integrity']['num_rendered_mismatches_vs_serial'])"; done | tee $O/speed.txt

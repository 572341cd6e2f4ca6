"""Randomised HIP-vs-oracle parity sweep over image shapes / sizes the unit tests do not enumerate
(tall images -> row-major span partition, more than 64 tile columns, tiny images, few Gaussians)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "semantic-gaussians_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from helpers import small_scene
from oracle import oracle as orc
import test_parity_gpu as T

rng = np.random.default_rng(1)
cases = [(3000, 128, 1100, 48), (3000, 128, 48, 1100), (500, 256, 2100, 16), (500, 128, 16, 2100), (7, 128, 64, 64),
         (1, 128, 16, 16), (20000, 128, 333, 257), (4000, 384, 257, 333), (2000, 128, 1296, 80), (2000, 128, 80, 1296)]
for _ in range(14):
    cases.append((int(rng.integers(1, 6000)), int(rng.choice([128, 256])), int(rng.integers(1, 700)), int(rng.integers(1, 500))))
bad = 0
for i, (P, C, W, H) in enumerate(cases):
    fx = float(max(W, H)) * 0.9
    scene, cam = small_scene(P=P, C=C, W=W, H=H, fx=fx, seed=100 + i)
    if i % 3 == 0:
        scene = scene._replace(scales=scene.scales * 4.0, opacities=scene.opacities * 0.3)
    try:
        for mode in (0, 2):
            T._check_forward(orc, scene, cam, binning_mode=mode)
        T._check_forward(orc, scene, cam, variant=15)
        print(f"ok   P={P} C={C} {W}x{H}")
    except AssertionError as e:
        bad += 1
        print(f"FAIL P={P} C={C} {W}x{H}: {str(e)[:200]}")
print("failures:", bad)
sys.exit(1 if bad else 0)

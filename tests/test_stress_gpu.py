"""Randomised HIP-vs-oracle parity sweep over image shapes / sizes the other tests do not enumerate:
tall images (the span partition then runs over rows first), more than 64 tile columns, tiny images, a
handful of Gaussians.  Every case checks binning modes 0 and 2 and the exact arithmetic."""
import numpy as np
import pytest

from helpers import has_experiments, small_scene
from test_parity_gpu import _check_forward

pytestmark = pytest.mark.gpu

_rng = np.random.default_rng(1)
CASES = [(3000, 128, 1100, 48), (3000, 128, 48, 1100), (500, 256, 2100, 16), (500, 128, 16, 2100), (7, 128, 64, 64),
         (1, 128, 16, 16), (20000, 128, 333, 257), (4000, 384, 257, 333), (2000, 128, 1296, 80), (2000, 128, 80, 1296)]
CASES += [(int(_rng.integers(1, 6000)), int(_rng.choice([128, 256])), int(_rng.integers(1, 700)), int(_rng.integers(1, 500)))
          for _ in range(14)]


@pytest.mark.parametrize("i", range(len(CASES)))
def test_random_shapes(orc, i):
    P, C, W, H = CASES[i]
    scene, cam = small_scene(P=P, C=C, W=W, H=H, fx=float(max(W, H)) * 0.9, seed=100 + i)
    if i % 3 == 0:
        scene = scene._replace(scales=scene.scales * 4.0, opacities=scene.opacities * 0.3)
    for mode in (0, 2) if has_experiments() else (0,):   # (mode 2 = round 1's tile-key sort on the library radix sort: make EXPERIMENTS=1)
        _check_forward(orc, scene, cam, binning_mode=mode)
    _check_forward(orc, scene, cam, variant=15)


@pytest.mark.gpu
@pytest.mark.parametrize("H", [16400, 32768])   # (32768 = 2048 tile rows: the longest axis the span partitions take)
def test_sweep_plan_many_segments(H):
    """The sweep's workgroup order (sweep_plan_kernel): more than 1024 segments take the bitonic sort, more than 4096
    the unsorted deal -- the feature map must be the row-major order's, bit for bit (variant 0x111004: the default kernels, row-major)."""
    import torch
    from test_parity_gpu import _hip_forward
    scene, cam = small_scene(P=30000, C=128, W=272, H=H, fx=300.0, seed=H)
    a = _hip_forward(scene, cam, variant=0x111004)
    ref_n, ref = a[0], a[1].clone()
    del a
    b = _hip_forward(scene, cam, variant=0)
    assert b[0] == ref_n and ref_n > 0
    assert torch.equal(b[1], ref)

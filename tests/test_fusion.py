"""the fusion mapper (dataset/fusion_utils.py:16-78) and the per-view accumulation
(fusion.py:139-147).

CPU: the NumPy oracle against outputs of the REFERENCE class itself (tests/golden/fusion_mapping.npz, made by
tests/golden/gen_fusion_fixtures.py in the build container).  GPU: the HIP kernels through the C-ABI against
those fixtures and against the oracle on larger seeded cases; bit-exact on the integer mapping and the fp32
sums, 1e-13 relative on the float64 weight (the device exp() is not glibc's)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oracle import fusion_oracle as fo  # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "fusion_mapping.npz"))
CASES = ("nodepth", "depthmap", "depthmap_tight", "surface")
DEV = "cuda:0"


def _case(name):
    depth = GOLD[name + "_depth"] if name + "_depth" in GOLD else ("surface" if name == "surface" else None)
    return (tuple(int(v) for v in GOLD[name + "_dim"]), int(GOLD[name + "_cut"]), float(GOLD[name + "_thres"]),
            GOLD[name + "_intr_in"], GOLD[name + "_wvt"], GOLD[name + "_coords"], depth, GOLD[name + "_mapping"],
            GOLD[name + "_weight"])


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_the_reference_mapper(name):
    dim, cut, thres, intr, wvt, coords, depth, want_map, want_w = _case(name)
    adj = fo.adjust_intrinsics(intr, dim)
    assert np.array_equal(adj, GOLD[name + "_intr_adj"])
    m, w = fo.compute_mapping(wvt, coords, dim, adj, cut, thres, depth)
    assert want_map[:, 2].sum() > 100
    assert np.array_equal(m, want_map)
    assert np.array_equal(w, want_w)


def _random_case(seed, N, W, H, mode):
    rng = np.random.default_rng(seed)
    intr = np.array([[W * rng.uniform(0.7, 1.3), 0.0, W * rng.uniform(0.45, 0.55)],
                     [0.0, W * rng.uniform(0.7, 1.3), H * rng.uniform(0.45, 0.55)], [0.0, 0.0, 1.0]])
    a = rng.uniform(-0.5, 0.5, size=3)
    Rx = np.array([[1, 0, 0], [0, np.cos(a[0]), -np.sin(a[0])], [0, np.sin(a[0]), np.cos(a[0])]])
    Ry = np.array([[np.cos(a[1]), 0, np.sin(a[1])], [0, 1, 0], [-np.sin(a[1]), 0, np.cos(a[1])]])
    w2c = np.eye(4)
    w2c[:3, :3] = Ry @ Rx
    w2c[:3, 3] = rng.uniform(-0.5, 0.5, size=3) + np.array([0.0, 0.0, 3.0])
    wvt = w2c.T.astype(np.float32)
    coords = (rng.normal(size=(N, 3)) * np.array([2.5, 2.0, 1.5])).astype(np.float32)
    coords[::211] = 0.0
    coords[::211, 2] = -w2c[2, 3]                       # z == 0 in camera space: division by zero
    depth = None
    if mode == "map":
        depth = rng.uniform(1.5, 4.5, size=(H, W)).astype(np.float32)
    elif mode == "surface":
        depth = "surface"
    return intr, wvt, coords, depth


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_mapper_reproduces_the_reference(name):
    from sgs_hip.fusion import PointCloudToImageMapper
    dim, cut, thres, intr, wvt, coords, depth, want_map, want_w = _case(name)
    mapper = PointCloudToImageMapper(dim, visibility_threshold=thres, cut_bound=cut, intrinsics=intr, device=DEV)
    assert np.array_equal(mapper.intrinsics, GOLD[name + "_intr_adj"])
    m, w = mapper.compute_mapping(wvt, coords, depth)    # the reference's call (NumPy in / out)
    assert m.dtype == np.int64 and m.shape == want_map.shape
    assert np.array_equal(m, want_map)
    assert np.allclose(w, want_w, rtol=1e-13, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,N,W,H,cut", [("none", 200_000, 648, 484, 10), ("map", 200_000, 648, 484, 10),
                                            ("surface", 300_000, 320, 240, 0), ("map", 1, 17, 9, 0),
                                            ("surface", 5, 16, 16, 3)])
def test_hip_mapper_matches_oracle_on_large_cases(mode, N, W, H, cut):
    """fusion_scannet.yaml's image size; points with z == 0 (inf / NaN projection -> the x86 "integer
    indefinite"), points behind the camera, several points per pixel in the z-buffer."""
    from sgs_hip.fusion import PointCloudToImageMapper
    intr, wvt, coords, depth = _random_case(N + W, N, W, H, mode)
    mapper = PointCloudToImageMapper((W, H), visibility_threshold=0.25, cut_bound=cut, intrinsics=intr, device=DEV)
    want_map, want_w = fo.compute_mapping(wvt, coords, (W, H), fo.adjust_intrinsics(intr, (W, H)), cut, 0.25, depth)
    m, w = mapper.compute_mapping_device(torch.from_numpy(wvt).to(DEV), torch.from_numpy(coords).to(DEV),
                                         depth if not isinstance(depth, np.ndarray) else torch.from_numpy(depth).to(DEV))
    assert np.array_equal(m.cpu().numpy(), want_map)
    assert np.allclose(w.cpu().numpy(), want_w, rtol=1e-13, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("C", [512, 768, 21, 6])
def test_hip_accumulate_is_the_reference_scatter(C):
    """Three views accumulated on the device against fusion.py:139-147 evaluated with torch on the host."""
    from sgs_hip.fusion import PointCloudToImageMapper, accumulate_features
    N, W, H = 20_000, 96, 64
    g = torch.Generator().manual_seed(C)
    feat_sum = torch.zeros(N, C, device=DEV)
    times = torch.zeros(N, 1, device=DEV)
    ref_sum, ref_times = torch.zeros(N, C), torch.zeros(N, 1)
    for view in range(3):
        intr, wvt, coords, depth = _random_case(100 + view, N, W, H, "map")
        if view == 0:
            xyz = coords
        mapper = PointCloudToImageMapper((W, H), intrinsics=intr, device=DEV)
        mapping, _ = mapper.compute_mapping_device(wvt, xyz, depth)
        features = torch.randn(C, H, W, generator=g)
        accumulate_features(feat_sum, times, features.to(DEV), mapping)
        mp = mapping.cpu()
        mask_k = mp[:, 2] != 0
        assert int(mask_k.sum()) > 500
        ref_times[mask_k] += 1
        ref_sum[mask_k] += features[:, mp[:, 0], mp[:, 1]].permute(1, 0)[mask_k]
    assert torch.equal(times.cpu(), ref_times)
    assert torch.equal(feat_sum.cpu(), ref_sum)
    # and the oracle's accumulate on the last view (channel-last entry point)
    fs, ts = np.zeros((N, C), np.float32), np.zeros(N, np.float32)
    fo.accumulate(fs, ts, features.numpy(), mp.numpy())
    d_sum, d_times = torch.zeros(N, C, device=DEV), torch.zeros(N, device=DEV)
    accumulate_features(d_sum, d_times, features.permute(1, 2, 0).contiguous().to(DEV), mapping, channel_last=True)
    assert np.array_equal(d_sum.cpu().numpy(), fs) and np.array_equal(d_times.cpu().numpy(), ts)

"""Generates tests/golden/fusion_mapping.npz by RUNNING the reference's PointCloudToImageMapper
(dataset/fusion_utils.py:16-78) in the build container (/root/reference does not exist on the GPU box).

The reference module does `from collections import Sequence`, which Python >= 3.10 no longer has; the
alias below (collections.Sequence = collections.abc.Sequence) is the whole accommodation -- the class
under test is executed unmodified.  Only inputs and outputs are stored.

    python tests/golden/gen_fusion_fixtures.py
"""
import collections
import collections.abc
import os
import sys

import numpy as np

collections.Sequence = collections.abc.Sequence   # py3.10: the name the reference imports
sys.path.insert(0, "/root/reference")
from dataset.fusion_utils import PointCloudToImageMapper  # noqa: E402

rng = np.random.default_rng(7)
out = {}
cases = [
    # name, image_dim (W,H), cut_bound, vis_thres, depth mode, N
    ("nodepth", (64, 48), 0, 0.25, "none", 3000),
    ("depthmap", (80, 60), 2, 0.25, "map", 4000),
    ("depthmap_tight", (64, 48), 0, 0.05, "map", 4000),
    ("surface", (48, 40), 1, 0.25, "surface", 3000),
]
for name, dim, cut, thres, mode, N in cases:
    W, H = dim
    # unadjusted intrinsics as a dataset would give them (principal point off the image centre on purpose)
    intr = np.array([[W * 0.9, 0.0, W * 0.47], [0.0, W * 0.95, H * 0.52], [0.0, 0.0, 1.0]])
    mapper = PointCloudToImageMapper(dim, visibility_threshold=thres, cut_bound=cut, intrinsics=intr)
    # a camera pose in the reference's convention: world_view_transform is the TRANSPOSED world-to-camera
    a = rng.uniform(-0.4, 0.4, size=3)
    Rx = np.array([[1, 0, 0], [0, np.cos(a[0]), -np.sin(a[0])], [0, np.sin(a[0]), np.cos(a[0])]])
    Ry = np.array([[np.cos(a[1]), 0, np.sin(a[1])], [0, 1, 0], [-np.sin(a[1]), 0, np.cos(a[1])]])
    Rz = np.array([[np.cos(a[2]), -np.sin(a[2]), 0], [np.sin(a[2]), np.cos(a[2]), 0], [0, 0, 1]])
    w2c = np.eye(4)
    w2c[:3, :3] = Rz @ Ry @ Rx
    w2c[:3, 3] = rng.uniform(-0.3, 0.3, size=3) + np.array([0.0, 0.0, 2.5])
    wvt = w2c.T.astype(np.float32)                       # what view.world_view_transform.cpu().numpy() holds
    coords = (rng.normal(size=(N, 3)) * np.array([1.6, 1.2, 1.0])).astype(np.float32)
    coords[::97, 2] = -2.5 - w2c[2, 3] + 2.5             # a few points near / behind the camera plane
    if mode == "map":
        # a plausible rendered depth: the z-buffer of the points, blurred by noise, with holes
        pc = (w2c @ np.concatenate([coords.astype(np.float64), np.ones((N, 1))], axis=1).T)
        depth = np.full((H, W), 2.5, dtype=np.float32)
        fxa, fya = mapper.intrinsics[0, 0], mapper.intrinsics[1, 1]
        u = np.round(pc[0] * fxa / pc[2] + W / 2).astype(int)
        v = np.round(pc[1] * fya / pc[2] + H / 2).astype(int)
        ok = (pc[2] > 0.2) & (u >= 0) & (u < W) & (v >= 0) & (v < H)
        for i in np.nonzero(ok)[0]:
            depth[v[i], u[i]] = min(depth[v[i], u[i]], pc[2, i])
        depth = (depth * rng.uniform(0.9, 1.1, size=depth.shape)).astype(np.float32)
        depth_arg = depth
    elif mode == "surface":
        depth, depth_arg = None, "surface"
    else:
        depth, depth_arg = None, None
    mapping, weight = mapper.compute_mapping(wvt, coords, depth_arg)
    out[name + "_dim"] = np.array(dim)
    out[name + "_cut"] = np.array(cut)
    out[name + "_thres"] = np.array(thres)
    out[name + "_intr_in"] = intr
    out[name + "_intr_adj"] = mapper.intrinsics
    out[name + "_wvt"] = wvt
    out[name + "_coords"] = coords
    if depth is not None:
        out[name + "_depth"] = depth
    out[name + "_mapping"] = mapping.astype(np.int64)
    out[name + "_weight"] = weight.astype(np.float64)
    print(name, "visible", int(mapping[:, 2].sum()), "of", N)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fusion_mapping.npz"), **out)

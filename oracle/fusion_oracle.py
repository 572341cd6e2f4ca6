"""CPU oracle of the 2-D -> 3-D fusion step.  TEST INFRASTRUCTURE ONLY: nothing in the
product path imports this file.

Restates, in float64 numpy,
  * PointCloudToImageMapper.__init__ / compute_mapping   (dataset/fusion_utils.py:16-78)
  * the per-view accumulation of fuse_one_scene           (fusion.py:127-147)
Pinned against outputs of the reference class itself: tests/golden/fusion_mapping.npz
(tests/golden/gen_fusion_fixtures.py runs the reference in the build container).
"""
import numpy as np

INT_MIN = np.iinfo(np.int64).min


def adjust_intrinsics(intrinsics, image_dim):
    """fusion_utils.py:22-28: focal lengths rescaled by image size / (2 * principal point), principal
    point moved to the image centre."""
    k = np.array(intrinsics, dtype=np.float64).copy()
    w, h = image_dim
    k[0, 0] *= w / (k[0, 2] * 2)
    k[1, 1] *= h / (k[1, 2] * 2)
    k[0, 2] = w / 2
    k[1, 2] = h / 2
    return k


def _round_to_int(v):
    """np.round(v).astype(int) as x86 does it: non-finite / out-of-range -> INT64_MIN."""
    r = np.rint(v)
    bad = ~np.isfinite(r) | (np.abs(r) >= 2.0 ** 63)
    out = np.where(bad, 0.0, r).astype(np.int64)
    out[bad] = INT_MIN
    return out


def compute_mapping(world_view_transform, coords, image_dim, intrinsics_adj, cut_bound=0, vis_thres=0.25,
                    depth=None):
    """fusion_utils.py:30-78.  world_view_transform: (4,4), the TRANSPOSED world-to-camera matrix (what
    view.world_view_transform holds); coords (N,3); depth: None | (H,W) array | "surface".
    Returns mapping (N,3) int64 rows (y, x, 1) or (0,0,0), weight (N,) float64."""
    w, h = image_dim
    m = np.asarray(world_view_transform, dtype=np.float64)
    c = np.asarray(coords, dtype=np.float64)
    n = c.shape[0]
    # camera-space point = (transform^T) @ (x, y, z, 1): an fma chain over the four terms in order
    cam = np.empty((3, n))
    for r in range(3):
        cam[r] = m[0, r] * c[:, 0] + m[1, r] * c[:, 1] + m[2, r] * c[:, 2] + m[3, r]
    with np.errstate(divide="ignore", invalid="ignore"):
        u = cam[0] * intrinsics_adj[0][0] / cam[2] + intrinsics_adj[0][2]
        v = cam[1] * intrinsics_adj[1][1] / cam[2] + intrinsics_adj[1][2]
    ui, vi = _round_to_int(u), _round_to_int(v)
    with np.errstate(over="ignore"):
        dist = np.sqrt((ui.astype(np.float64) - w / 2) ** 2 + (vi.astype(np.float64) - h / 2) ** 2)
    inside = (ui >= cut_bound) & (vi >= cut_bound) & (ui < w - cut_bound) & (vi < h - cut_bound)
    if isinstance(depth, str):   # "surface": z-buffer of the points themselves (order independent: a min)
        zbuf = np.full((h, w), 999999.0)
        sel = inside & (cam[2] > 0.2)
        np.minimum.at(zbuf, (vi[sel], ui[sel]), cam[2][sel])
        depth = zbuf
    if depth is not None:
        d = np.asarray(depth, dtype=np.float64)
        idx = np.nonzero(inside)[0]
        dcur = d[vi[idx], ui[idx]]
        vis = np.abs(dcur - cam[2][idx]) <= vis_thres * dcur
        inside = np.zeros(n, dtype=bool)
        inside[idx[vis]] = True
    else:
        inside = inside & (cam[2] > 0)
    mapping = np.zeros((n, 3), dtype=np.int64)
    mapping[inside, 0] = vi[inside]
    mapping[inside, 1] = ui[inside]
    mapping[inside, 2] = 1
    return mapping, np.exp(-dist / 10)


def accumulate(feat_sum, times, features_chw, mapping):
    """fusion.py:139-147 for one view: visible points add the feature vector of their pixel and count it."""
    vis = mapping[:, 2] != 0
    feat_sum[vis] += features_chw[:, mapping[vis, 0], mapping[vis, 1]].T
    times[vis] += 1
    return feat_sum, times

"""Pinhole camera matrices in the convention the rasteriser expects.

The rasteriser consumes the *transposed* world->view and full projection
matrices, flattened row-major (element m[4*col+row] of the column-vector
convention).  Input generator for the synthetic benchmarks and tests -- the
reference builds the same matrices in scene/camera.py:81-94 with the helpers of
utils/graphics_utils.py; the outputs here are pinned bit for bit against fixtures
generated from those helpers (tests/golden/reference_fixtures.json,
tests/test_oracle_kat.py::test_cameras_match_reference_fixtures).
"""
import math
from typing import NamedTuple

import numpy as np
import torch


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


def world2view(R, t, translate=(0.0, 0.0, 0.0), scale=1.0):
    """The 4 x 4 world -> view matrix of a camera given the way the reference's loaders give it: R = the camera-to-world rotation, t = the
    world-to-view translation, i.e.  x_view = R^T x_world + t  (same convention and outputs as utils/graphics_utils.py:38-50; pinned
    by tests/golden/reference_fixtures.json).  `translate` / `scale` move the camera CENTRE, c -> (c + translate) * scale (the
    reference's scene normalisation); the rotation is untouched, so only the last column changes: t' = -R^T c'."""
    Rm = np.asarray(R, dtype=np.float64)
    centre = -Rm @ np.asarray(t, dtype=np.float64)                      # x_view = 0  <=>  x_world = -R t
    centre = (centre + np.asarray(translate, dtype=np.float64)) * scale
    m = np.eye(4)
    m[:3, :3] = Rm.T
    m[:3, 3] = -Rm.T @ centre
    return m.astype(np.float32)


def projection_matrix(znear, zfar, fovX, fovY):
    """Perspective projection of a symmetric frustum, z_view in [znear, zfar] -> depth in [0, 1], w = z_view (the reference's
    utils/graphics_utils.py:53-78 reduces to this for its symmetric left / right, top / bottom): five non-zero entries."""
    P = torch.zeros(4, 4)
    P[0, 0] = 1.0 / math.tan(0.5 * fovX)
    P[1, 1] = 1.0 / math.tan(0.5 * fovY)
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    P[3, 2] = 1.0
    return P


class PinholeCamera(NamedTuple):
    image_width: int
    image_height: int
    FoVx: float
    FoVy: float
    world_view_transform: torch.Tensor  # (4,4) transposed W2C
    full_proj_transform: torch.Tensor   # (4,4) transposed (P @ W2C)
    camera_center: torch.Tensor         # (3,)

    @property
    def tanfovx(self):
        return math.tan(self.FoVx * 0.5)

    @property
    def tanfovy(self):
        return math.tan(self.FoVy * 0.5)

    def to(self, device):
        return self._replace(
            world_view_transform=self.world_view_transform.to(device),
            full_proj_transform=self.full_proj_transform.to(device),
            camera_center=self.camera_center.to(device))


def make_camera(R, T, FoVx, FoVy, width, height, znear=0.01, zfar=100.0,
                trans=(0.0, 0.0, 0.0), scale=1.0):
    """scene/camera.py:81-94."""
    # the rasteriser takes both matrices TRANSPOSED (row-vector convention: x_clip = x_world . full)
    wvt = torch.from_numpy(world2view(R, T, trans, scale)).t()
    full = wvt @ projection_matrix(znear, zfar, FoVx, FoVy).t()
    center = wvt.inverse()[3, :3]
    return PinholeCamera(int(width), int(height), float(FoVx), float(FoVy), wvt.contiguous(),
                         full.contiguous(), center.contiguous())


def pinhole(width, height, fx, fy=None, R=None, T=None):
    fy = fx if fy is None else fy
    R = np.eye(3) if R is None else R
    T = np.zeros(3) if T is None else T
    return make_camera(R, T, focal2fov(fx, width), focal2fov(fy, height), width, height)


def ring_cameras(n, radius, width, height, fx, look_at=(0.0, 0.0, 0.0)):
    """n cameras on a horizontal ring looking at `look_at`."""
    cams = []
    for i in range(n):
        a = 2 * math.pi * i / n
        c = np.array([radius * math.sin(a), 0.0, -radius * math.cos(a)]) + np.asarray(look_at)
        fwd = np.asarray(look_at) - c
        fwd = fwd / np.linalg.norm(fwd)
        up = np.array([0.0, -1.0, 0.0])
        right = np.cross(up, fwd)
        right /= np.linalg.norm(right)
        dn = np.cross(fwd, right)
        C2W_R = np.stack([right, dn, fwd], axis=1)  # columns = camera axes in world
        W2C_R = C2W_R.T
        T = -W2C_R @ c
        # the reference's Camera takes R = C2W rotation and T = W2C translation
        cams.append(make_camera(C2W_R, T, focal2fov(fx, width), focal2fov(fx, height), width, height))
    return cams

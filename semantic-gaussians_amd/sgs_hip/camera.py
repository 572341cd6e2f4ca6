"""Pinhole camera matrices in the convention the rasteriser expects.

The rasteriser consumes the *transposed* world->view and full projection
matrices, flattened row-major (element m[4*col+row] of the column-vector
convention).  This restates how the reference builds them
(scene/camera.py:81-94, utils/graphics_utils.py:38-78) so that synthetic
benchmarks and tests can run without the reference's scene loaders; the
values are pinned against fixtures generated from the reference's own helpers
(tests/golden/cameras.json).
"""
import math
from typing import NamedTuple

import numpy as np
import torch


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


def world2view(R, t, translate=(0.0, 0.0, 0.0), scale=1.0):
    """utils/graphics_utils.py:38-50 (getWorld2View2)."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = np.asarray(R, dtype=np.float64).transpose()
    Rt[:3, 3] = np.asarray(t, dtype=np.float64)
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    cam_center = C2W[:3, 3]
    cam_center = (cam_center + np.asarray(translate, dtype=np.float64)) * scale
    C2W[:3, 3] = cam_center
    Rt = np.linalg.inv(C2W)
    return np.float32(Rt)


def projection_matrix(znear, zfar, fovX, fovY):
    """utils/graphics_utils.py:53-78 (getProjectionMatrix)."""
    tanHalfFovY = math.tan(fovY / 2)
    tanHalfFovX = math.tan(fovX / 2)
    top = tanHalfFovY * znear
    bottom = -top
    right = tanHalfFovX * znear
    left = -right
    P = torch.zeros(4, 4)
    z_sign = 1.0
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = z_sign
    P[2, 2] = z_sign * zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
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
    wvt = torch.tensor(world2view(R, T, trans, scale)).transpose(0, 1)
    proj = projection_matrix(znear=znear, zfar=zfar, fovX=FoVx, fovY=FoVy).transpose(0, 1)
    full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
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

"""Generates tests/golden/reference_fixtures.json by IMPORTING the reference's own Python
helpers (run in the build container only; /root/reference does not exist on the GPU box).

What the reference can pin for this path: the camera-matrix convention
(scene/camera.py:81-94 + utils/graphics_utils.py) and the SH polynomial (utils/sh_utils.py
eval_sh == the kernel's SH->RGB before +0.5 and clamp).  Only inputs and expected outputs are
stored -- no reference source text.

    python tests/golden/gen_reference_fixtures.py
"""
import importlib.util
import json
import math
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
from utils.graphics_utils import getWorld2View2, getProjectionMatrix, focal2fov  # noqa: E402
from utils.sh_utils import eval_sh  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_camera", os.path.join(REF, "scene", "camera.py"))
ref_camera = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_camera)

out = {"cameras": [], "sh": []}
rng = np.random.default_rng(0)
cases = [
    (np.eye(3), np.array([0.0, 0.0, 3.0]), 500.0, 500.0, 640, 480),
    (np.eye(3), np.array([0.0, 0.0, 0.0]), 1170.0, 1170.0, 1296, 968),
]
for _ in range(4):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    T = rng.normal(size=3) * 2
    fx, fy = float(rng.uniform(200, 1500)), float(rng.uniform(200, 1500))
    cases.append((R, T, fx, fy, int(rng.integers(100, 1400)), int(rng.integers(100, 1000))))
for R, T, fx, fy, W, H in cases:
    fovx, fovy = focal2fov(fx, W), focal2fov(fy, H)
    cam = ref_camera.Camera(colmap_id=0, R=R, T=T, FoVx=fovx, FoVy=fovy,
                            image=torch.zeros(3, H, W), gt_alpha_mask=None, image_name="x", image_path="",
                            uid=0, device="cpu")
    out["cameras"].append(dict(
        R=R.tolist(), T=T.tolist(), fx=fx, fy=fy, W=W, H=H, FoVx=fovx, FoVy=fovy,
        world_view_transform=cam.world_view_transform.numpy().astype(np.float64).tolist(),
        full_proj_transform=cam.full_proj_transform.numpy().astype(np.float64).tolist(),
        camera_center=cam.camera_center.numpy().astype(np.float64).tolist()))

g = torch.Generator().manual_seed(1)
for deg in range(4):
    n = 16
    sh = torch.randn(n, 3, 16, generator=g)          # eval_sh layout: (..., C, coeffs)
    d = torch.randn(n, 3, generator=g)
    d = d / d.norm(dim=1, keepdim=True)
    res = eval_sh(deg, sh, d)                          # (n,3)
    out["sh"].append(dict(deg=deg, sh=sh.numpy().astype(np.float64).tolist(),
                          dirs=d.numpy().astype(np.float64).tolist(),
                          result=res.numpy().astype(np.float64).tolist()))

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_fixtures.json"), "w") as f:
    json.dump(out, f)
print("cameras", len(out["cameras"]), "sh", len(out["sh"]))

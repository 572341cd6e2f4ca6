"""Generates tests/golden/projection_reference.npz by IMPORTING the reference's own Python projection helper (run in the build
container only; /root/reference does not exist on the GPU box).

utils/graphics_utils.py:24-31 (geom_transform_points) is the Python twin of the kernels' projection: homogeneous point x the (transposed,
row-major) matrix, then a division by (w + 0.0000001) -- the `p_w = 1 / (p_hom.w + 0.0000001f)` of CR/cuda_rasterizer/forward.cu:199-200.
For each fixture camera (the same six as reference_fixtures.json, built by the reference's Camera class) it is applied to a few thousand
points with the full projection (-> NDC x, y, z) and with the view matrix (-> view-space z, the kernels' depth).  Only inputs and
expected outputs are stored -- no reference source text.

    python tests/golden/gen_projection_fixture.py
"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
from utils.graphics_utils import geom_transform_points  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_camera", os.path.join(REF, "scene", "camera.py"))
ref_camera = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_camera)

HERE = os.path.dirname(os.path.abspath(__file__))
fx = json.load(open(os.path.join(HERE, "reference_fixtures.json")))
g = torch.Generator().manual_seed(11)
out = {}
for ci, c in enumerate(fx["cameras"]):
    cam = ref_camera.Camera(colmap_id=0, R=np.array(c["R"]), T=np.array(c["T"]), FoVx=c["FoVx"], FoVy=c["FoVy"],
                            image=torch.zeros(3, c["H"], c["W"]), gt_alpha_mask=None, image_name="x", image_path="",
                            uid=0, device="cpu")
    # points in the camera's frustum slab (view space), mapped back to the world: p_view = p_world R + T  (row vectors, W2C^T)
    n = 2048
    z = torch.rand(n, generator=g) * 7.5 + 0.25
    tx, ty = np.tan(c["FoVx"] / 2), np.tan(c["FoVy"] / 2)
    x = (torch.rand(n, generator=g) * 2 - 1) * z * tx * 1.15        # a margin outside the image, like the bench generator
    y = (torch.rand(n, generator=g) * 2 - 1) * z * ty * 1.15
    pv = torch.stack([x, y, z], 1).double()
    R, T = torch.tensor(c["R"]).double(), torch.tensor(c["T"]).double()
    pw = ((pv - T) @ R.T).float()                                   # W2C = [R^T | T]  =>  p_world = R (p_view - T)
    ndc = geom_transform_points(pw, cam.full_proj_transform)
    view = geom_transform_points(pw, cam.world_view_transform)
    out[f"points_{ci}"] = pw.numpy()
    out[f"ndc_{ci}"] = ndc.numpy()
    out[f"view_{ci}"] = view.numpy()
np.savez_compressed(os.path.join(HERE, "projection_reference.npz"), **out)
print({k: v.shape for k, v in out.items() if k.endswith("_0")}, "cameras", len(fx["cameras"]))

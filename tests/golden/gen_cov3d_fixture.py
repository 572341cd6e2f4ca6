"""Generates tests/golden/cov3d_reference.npz by IMPORTING the reference's own Python route to the 3-D covariance
(model/gaussian_model.py:34-38 build_covariance_from_scaling_rotation = strip_symmetric(L L^T), L = build_scaling_rotation(modifier * scale, q),
utils/general_utils.py:66-115) -- what `pipe.compute_cov3D_python` feeds the rasteriser as cov3D_precomp and what the kernel's own
computeCov3D (CR/cuda_rasterizer/forward.cu:115-150) must agree with.  The helpers allocate on device="cuda"; there is no GPU in the build
container, so the allocations are redirected to the CPU for the duration of the calls (same arithmetic, fp32).  Only inputs and
expected outputs are stored -- no reference source text.  Run in the build container only (/root/reference does not exist on the GPU box):

    python tests/golden/gen_cov3d_fixture.py
"""
import os
import sys
from unittest import mock

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
from utils.general_utils import build_scaling_rotation, strip_symmetric  # noqa: E402

_zeros = torch.zeros


def _zeros_on_cpu(*a, **k):
    k.pop("device", None)
    return _zeros(*a, **k)


g = torch.Generator().manual_seed(7)
n = 512
log_scale = torch.randn(n, 3, generator=g) * 1.5 - 3.0          # scales over four decades, anisotropic
scales = torch.exp(log_scale)
q = torch.randn(n, 4, generator=g)
q = q / q.norm(dim=1, keepdim=True)                              # as GaussianModel.get_rotation hands them over (normalised)
q[0] = torch.tensor([1.0, 0.0, 0.0, 0.0])                        # identity
q[1] = torch.tensor([0.0, 1.0, 0.0, 0.0])                        # half turn about x
modifiers = [1.0, 0.5, 2.25]
out = {"scales": scales.numpy(), "rotations": q.numpy(), "modifiers": np.asarray(modifiers, np.float32)}
with mock.patch.object(torch, "zeros", _zeros_on_cpu):
    for i, m in enumerate(modifiers):
        L = build_scaling_rotation(m * scales, q)
        out[f"cov3D_{i}"] = strip_symmetric(L @ L.transpose(1, 2)).numpy()
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cov3d_reference.npz"), **out)
print({k: v.shape for k, v in out.items()})

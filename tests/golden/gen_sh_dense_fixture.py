"""Generates tests/golden/sh_dense.npz by IMPORTING the reference's own utils/sh_utils.py eval_sh (run in the build container only;
/root/reference does not exist on the GPU box): 256 directions on the front hemisphere per SH degree 0-3, random coefficients, the
reference's polynomial evaluated in float64.  ADVICE r4: the kernel and the oracle both evaluate the GENERATED monomial table
(tools/gen_sh_table.py), so their bit-exact agreement no longer checks the table against the reference's hand-expanded polynomials
(CR/cuda_rasterizer/forward.cu:20-71 = eval_sh); this dense known-answer set does -- a wrong coefficient or a dropped term of any
basis function shows at 1e-3, the test bar is a few fp32 ulps.  Only inputs and expected outputs are stored.

    python tests/golden/gen_sh_dense_fixture.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from utils.sh_utils import eval_sh  # noqa: E402

g = torch.Generator().manual_seed(7)
out = {}
for deg in range(4):
    n = 256
    sh = torch.randn(n, 3, 16, generator=g).float()
    d = torch.randn(n, 3, generator=g).float()
    d[:, 2] = d[:, 2].abs() + 0.35                     # in front of a camera at the origin looking down +z
    pos = (d * (1.0 + 3.0 * torch.rand(n, 1, generator=g))).float()   # the Gaussian's position: the kernel normalises (pos - campos) itself
    dn = pos.double() / pos.double().norm(dim=1, keepdim=True)
    res = eval_sh(deg, sh.double(), dn)                # (n,3) float64, the reference's own code
    out[f"sh{deg}"] = sh.numpy()
    out[f"pos{deg}"] = pos.numpy()
    out[f"res{deg}"] = res.numpy()
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "sh_dense.npz"), **out)
print({k: v.shape for k, v in out.items()})

"""Frozen synthetic scene generator of BASELINE.md section 3 / SURVEY.md 8(d).

torch.Generator().manual_seed(seed), fp32, generated on CPU (draw order is part of
the definition: xyz, log-scales, rotations, opacity logits, features).
"""
import math
from typing import NamedTuple

import torch

from .camera import pinhole


class Scene(NamedTuple):
    means3D: torch.Tensor    # (P,3)
    scales: torch.Tensor     # (P,3)
    rotations: torch.Tensor  # (P,4) normalised (w,x,y,z)
    opacities: torch.Tensor  # (P,1)
    features: torch.Tensor   # (P,C) row-L2-normalised
    bg: torch.Tensor         # (C,)

    def to(self, device):
        return Scene(*[t.to(device) for t in self])


CONFIGS = {
    # name: (P, C, W, H, fx)
    "cfg1": (10_000, 3, 256, 256, 230.0),
    "cfg2": (500_000, 3, 1296, 968, 1170.0),
    "cfg2_half": (500_000, 3, 648, 484, 585.0),
    "cfg3": (1_000_000, 512, 1296, 968, 1170.0),
    "cfg4": (5_000_000, 768, 1297, 840, 1170.0),
    "cfg5": (50_000_000, 256, 1296, 968, 1170.0),
}


def make_scene(P, C, W, H, fx, seed=0, feature_chunk=1 << 18, features=True):
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(P, 3, generator=g)
    z = 0.5 + 5.5 * u[:, 2]
    hx = W / (2.0 * fx)
    hy = H / (2.0 * fx)
    x = (2.0 * u[:, 0] - 1.0) * 1.1 * hx * z
    y = (2.0 * u[:, 1] - 1.0) * 1.1 * hy * z
    means3D = torch.stack([x, y, z], dim=1).contiguous()
    scales = torch.exp(math.log(0.01) + 0.4 * torch.randn(P, 3, generator=g))
    rot = torch.randn(P, 4, generator=g)
    rot = rot / rot.norm(dim=1, keepdim=True)
    opac = torch.sigmoid(1.5 * torch.randn(P, 1, generator=g))
    if features:
        feats = torch.empty(P, C)
        for s in range(0, P, feature_chunk):
            e = min(P, s + feature_chunk)
            f = torch.randn(e - s, C, generator=g)
            feats[s:e] = f / f.norm(dim=1, keepdim=True)
    else:
        feats = torch.empty(0, C)
    return Scene(means3D, scales, rot, opac, feats, torch.zeros(C))


def make_config(name, seed=0, P=None, C=None, features=True):
    P0, C0, W, H, fx = CONFIGS[name]
    P = P0 if P is None else P
    C = C0 if C is None else C
    return make_scene(P, C, W, H, fx, seed, features=features), pinhole(W, H, fx)

"""Writes tests/golden/fusion_full.pt and fusion_masked.pt in the format the reference's fusion.py:234-257 writes -- with torch alone,
never touching sgs_hip.io: a dict {"feat": float16 (N, C), "mask_full": bool (P)} through torch.save, N = P for a scene below
n_split_points (all-true mask), N = mask.sum() otherwise (only the masked rows are stored, in index order).  The values are
exactly representable in fp16 so that the readers' float32 results are known in closed form: feat[i][c] = (i - 3) / 4 + c / 64.

    python tests/golden/gen_fusion_pt_fixture.py
"""
import os

import torch

here = os.path.dirname(os.path.abspath(__file__))
P, C = 12, 8
full = (torch.arange(P, dtype=torch.float32)[:, None] - 3.0) / 4.0 + torch.arange(C, dtype=torch.float32)[None, :] / 64.0
torch.save({"feat": full.cpu().half(), "mask_full": torch.ones(P, dtype=torch.bool)}, os.path.join(here, "fusion_full.pt"))
mask = torch.zeros(P, dtype=torch.bool)
mask[[1, 4, 5, 9, 11]] = True
torch.save({"feat": full[mask].cpu().half(), "mask_full": mask}, os.path.join(here, "fusion_masked.pt"))
print("written", P, C, int(mask.sum()))

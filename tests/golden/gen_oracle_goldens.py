"""Generates tests/golden/small_scene.npz: the INPUT tensors are stored (torch's CPU exp /
sigmoid / norm kernels are not bit-reproducible across host CPUs, so a seed is not enough),
expected outputs come from the CPU oracle.  Committed so that (a) the oracle itself is regression-pinned
and (b) the GPU parity tests have a fixed vector that does not depend on building the oracle.

    python tests/golden/gen_oracle_goldens.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "semantic-gaussians_amd"), os.path.dirname(HERE)):
    sys.path.insert(0, p)

from helpers import small_scene, oracle_forward  # noqa: E402
from oracle import oracle as orc  # noqa: E402

PARAMS = dict(P=900, C=6, W=80, H=48, fx=70.0, seed=42)


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    scene, cam = small_scene(**PARAMS)
    bg = np.linspace(-0.5, 0.5, PARAMS["C"]).astype(np.float32)
    fw = oracle_forward(orc, scene, cam, bg=bg, want_depth=False)
    fwd = oracle_forward(orc, scene, cam, colors=scene.features[:, :3], bg=bg[:3], want_depth=True)
    rng = np.random.default_rng(7)
    dL = rng.normal(size=fw["out"].shape).astype(np.float32)
    gr = orc.backward(fw, dL, scene.means3D.numpy(), cam.world_view_transform.numpy(),
                      cam.full_proj_transform.numpy(), cam.camera_center.numpy(), PARAMS["W"],
                      PARAMS["H"], cam.tanfovx, cam.tanfovy, bg, scales=scene.scales.numpy(),
                      rotations=scene.rotations.numpy())
    d2 = orc.dist2(scene.means3D.numpy())
    np.savez_compressed(
        os.path.join(HERE, "small_scene.npz"),
        params=np.array([PARAMS[k] for k in ("P", "C", "W", "H")], np.int64), fx=PARAMS["fx"],
        seed=PARAMS["seed"], bg=bg, dL=dL,
        in_means3D=scene.means3D.numpy(), in_scales=scene.scales.numpy(),
        in_rotations=scene.rotations.numpy(), in_opacities=scene.opacities.numpy(),
        in_features=scene.features.numpy(),
        num_rendered=fw["num_rendered"], radii=fw["radii"], ranges=fw["ranges"],
        keys_sorted_sha256=digest(fw["keys_sorted"]), point_list_sha256=digest(fw["point_list"]),
        keys_sorted_head=fw["keys_sorted"][:64], point_list_head=fw["point_list"][:64],
        out=fw["out"], final_T=fw["final_T"], n_contrib=fw["n_contrib"],
        rgb_out=fwd["out"], depth=fwd["depth"],
        dL_dmeans3D=gr["dL_dmeans3D"], dL_dcolors=gr["dL_dcolors"], dL_dopacity=gr["dL_dopacity"],
        dL_dscales=gr["dL_dscales"], dL_drotations=gr["dL_drotations"], dL_dmean2D=gr["dL_dmean2D"],
        dist2=d2)
    print("wrote small_scene.npz: L =", fw["num_rendered"])


if __name__ == "__main__":
    main()

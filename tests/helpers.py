"""Shared helpers for the test-suite: small seeded scenes in numpy/torch form."""
import numpy as np
import torch

from sgs_hip.camera import pinhole
from sgs_hip.synthetic import make_scene


def small_scene(P=2000, C=8, W=160, H=112, fx=150.0, seed=0):
    """Seeded scene following BASELINE.md's generator at a size the oracle finishes in
    well under a second."""
    scene = make_scene(P, C, W, H, fx, seed=seed)
    # denser / bigger splats than the headline generator so that tiles saturate and the
    # early-stop / n_contrib logic is exercised at this small P
    scene = scene._replace(scales=scene.scales * 6.0)
    cam = pinhole(W, H, fx)
    return scene, cam


def oracle_forward(orc, scene, cam, want_depth=False, colors=None, shs=None, sh_degree=0,
                   cov3D_precomp=None, bg=None, scale_modifier=1.0):
    C = 3 if (shs is not None) else (scene.features.shape[1] if colors is None else colors.shape[1])
    kw = {}
    if cov3D_precomp is None:
        kw.update(scales=scene.scales.numpy(), rotations=scene.rotations.numpy())
    else:
        kw.update(cov3D_precomp=cov3D_precomp)
    if shs is not None:
        kw.update(shs=shs, sh_degree=sh_degree)
    else:
        kw.update(colors_precomp=(scene.features if colors is None else colors).numpy())
    bg = scene.bg.numpy() if bg is None else bg
    return orc.forward(scene.means3D.numpy(), scene.opacities.numpy(),
                       cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(),
                       cam.camera_center.numpy(), cam.image_width, cam.image_height, cam.tanfovx,
                       cam.tanfovy, bg, C, scale_modifier=scale_modifier, want_depth=want_depth,
                       **kw)


def has_experiments():
    """The loaded libsgs_hip.so was built with `make EXPERIMENTS=1` (development forms of the blend kernels, csrc/Makefile)."""
    from sgs_hip import raster
    return bool(raster.build_flags() & 4)


def need_experiments(what="this blend variant"):
    import pytest
    if not has_experiments():
        pytest.skip(f"{what} is a development form built only by `make EXPERIMENTS=1` (the product library answers it with SGS_EINVAL)")

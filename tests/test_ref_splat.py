"""Pins the per-Gaussian math -- projection, EWA covariance, conic, SH colour and the whole backward
chain to the RAW inputs (CR/cuda_rasterizer/forward.cu:74-255, backward.cu:20-391) -- against the
independent float64 autograd splat of tests/ref_splat.py, on anisotropic, rotated, off-axis and
clamp-active Gaussians seen by a rotated, translated camera with fx != fy.

CPU part: the oracle (oracle/sgs_oracle.c).  GPU part: the HIP path through the C-ABI.
Both must agree with torch.autograd; neither shares source text with ref_splat.py.
"""
import json
import math
import os

import numpy as np
import pytest
import torch

import ref_splat
from sgs_hip.camera import make_camera, focal2fov

F64 = torch.float64
HERE = os.path.dirname(os.path.abspath(__file__))


def _rotation(axis, deg):
    axis = np.asarray(axis, np.float64)
    axis /= np.linalg.norm(axis)
    a = math.radians(deg)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + math.sin(a) * K + (1 - math.cos(a)) * K @ K


def pin_scene(seed=0, P=56, W=64, H=48, fx=58.0, fy=51.0, n_clamp=8):
    """Gaussians placed in VIEW space (so that the clamp-active ones are where we want them), moved to
    world space through a camera that is rotated 23 degrees about a skew axis and translated."""
    rng = np.random.default_rng(seed)
    Rc2w = _rotation([0.3, 1.0, -0.2], 23.0)          # the reference's Camera takes R = C2W rotation
    Tw2c = np.array([0.4, -0.25, 0.6])                # and T = W2C translation
    cam = make_camera(Rc2w, Tw2c, focal2fov(fx, W), focal2fov(fy, H), W, H)
    tanx, tany = cam.tanfovx, cam.tanfovy
    z = rng.uniform(1.0, 4.0, P)
    xr = rng.uniform(-1.0, 1.0, P) * tanx
    yr = rng.uniform(-1.0, 1.0, P) * tany
    scales = np.exp(rng.normal(math.log(0.08), 0.7, (P, 3)))       # anisotropic: ratios up to ~30x
    # clamp-active: |x/z| or |y/z| beyond 1.3 tan(fov/2); big enough to still reach the image
    k = np.arange(n_clamp)
    xr[k] = np.where(k % 2 == 0, 1.0, -1.0) * rng.uniform(1.35, 1.6, n_clamp) * tanx
    yr[k[n_clamp // 2:]] = rng.uniform(1.35, 1.5, n_clamp - n_clamp // 2) * tany
    scales[k] = rng.uniform(0.25, 0.6, (n_clamp, 3)) * z[k, None]
    view_pts = np.stack([xr * z, yr * z, z], 1)
    W2C_R = Rc2w.T
    world = (view_pts - Tw2c) @ W2C_R                               # p_w = R^T (p_v - t), as row vectors
    rot = rng.normal(size=(P, 4))
    rot /= np.linalg.norm(rot, axis=1, keepdims=True)
    opac = rng.uniform(0.2, 0.95, (P, 1))
    f32 = lambda a: torch.tensor(np.asarray(a, np.float32))
    return dict(cam=cam, means3D=f32(world), scales=f32(scales), rotations=f32(rot), opacities=f32(opac),
                W=W, H=H, rng=rng)


def _d(t):
    return t.detach().to(F64).clone().requires_grad_(True)


def _ref(sc, C, bg, dL, colors=None, shs=None, sh_degree=0, cov3D=None, mod=1.0):
    cam = sc["cam"]
    leaves = dict(means3D=_d(sc["means3D"]), opacities=_d(sc["opacities"]),
                  means2D=torch.zeros(sc["means3D"].shape[0], 2, dtype=F64, requires_grad=True))
    kw = {}
    if cov3D is None:
        leaves["scales"], leaves["rotations"] = _d(sc["scales"]), _d(sc["rotations"])
        kw.update(scales=leaves["scales"], rotations=leaves["rotations"], scale_modifier=mod)
    else:
        leaves["cov3D"] = _d(cov3D)
        kw.update(cov3D_precomp=leaves["cov3D"])
    if shs is not None:
        leaves["shs"] = _d(shs)
        kw.update(shs=leaves["shs"], sh_degree=sh_degree)
    else:
        leaves["colors"] = _d(colors)
        kw.update(colors_precomp=leaves["colors"])
    r = ref_splat.render(leaves["means3D"], leaves["opacities"], cam.world_view_transform.to(F64),
                         cam.full_proj_transform.to(F64), cam.camera_center.to(F64), sc["W"], sc["H"],
                         cam.tanfovx, cam.tanfovy, torch.tensor(bg, dtype=F64),
                         means2D_offset=leaves["means2D"], **kw)
    (r["out"] * torch.tensor(dL, dtype=F64)).sum().backward()
    return r, {k: v.grad.numpy() for k, v in leaves.items()}


def _oracle(orc, sc, C, bg, dL, colors=None, shs=None, sh_degree=0, cov3D=None, mod=1.0):
    cam = sc["cam"]
    kw = dict(scale_modifier=mod)
    if cov3D is None:
        kw.update(scales=sc["scales"].numpy(), rotations=sc["rotations"].numpy())
    else:
        kw.update(cov3D_precomp=cov3D.numpy())
    if shs is not None:
        kw.update(shs=shs.numpy(), sh_degree=sh_degree)
    else:
        kw.update(colors_precomp=colors.numpy())
    args = (cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), cam.camera_center.numpy(),
            sc["W"], sc["H"], cam.tanfovx, cam.tanfovy)
    fw = orc.forward(sc["means3D"].numpy(), sc["opacities"].numpy(), *args, bg, C, **kw)
    kw.pop("colors_precomp", None)
    g = orc.backward(fw, dL, sc["means3D"].numpy(), *args, bg, **kw)
    return fw, g


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


FWD_TOL = 2e-5      # per-Gaussian fp32 forward quantities vs float64 (relative to the largest entry)
GRAD_TOL = 2e-4     # backward vs autograd (fp32 kernels, float64 reference)


def _check_forward(fw, r):
    vis = r["vis"].numpy()
    assert np.array_equal(fw["radii"], r["radii"].numpy())
    assert vis.sum() >= 40
    assert _rel(fw["depths"][vis], r["depth"].detach().numpy()[vis]) < FWD_TOL
    assert _rel(fw["means2D"][vis], r["pix"].detach().numpy()[vis]) < FWD_TOL
    conic = r["conic"].detach().numpy()[vis]
    # entry-wise relative to each Gaussian's own largest conic entry
    err = np.abs(fw["conic_opacity"][vis, :3] - conic).max(1) / np.abs(conic).max(1)
    assert err.max() < 2e-4, err.max()     # Sigma2 is a difference of O(1e3) products in fp32
    out = r["out"].detach().numpy()
    assert np.abs(fw["out"] - out).max() < 2e-5 * max(1.0, np.abs(out).max())


def _check_grads(g, ref, names):
    for ours, theirs in names:
        want = ref[theirs]
        got = np.asarray(g[ours], np.float64).reshape(-1)[: want.size].reshape(want.shape) \
            if g[ours].size == want.size else np.asarray(g[ours], np.float64)[..., : want.shape[-1]]
        assert _rel(got, want) < GRAD_TOL, (ours, _rel(got, want))
        assert np.abs(want).max() > 0, ours


GRADS_SR = [("dL_dmeans3D", "means3D"), ("dL_dscales", "scales"), ("dL_drotations", "rotations"),
            ("dL_dopacity", "opacities"), ("dL_dmean2D", "means2D")]


def _case(sc, kind):
    rng, P, W, H = sc["rng"], sc["means3D"].shape[0], sc["W"], sc["H"]
    if kind == "precomp":
        C = 5
        colors = torch.tensor(rng.normal(size=(P, C)).astype(np.float32))
        return C, dict(colors=colors), GRADS_SR + [("dL_dcolors", "colors")]
    if kind == "sh":
        shs = torch.tensor((rng.normal(size=(P, 16, 3)) * 0.4).astype(np.float32))
        shs[:, 0, :] -= 0.6          # a good share of negative (clamped) colours
        return 3, dict(shs=shs, sh_degree=3), GRADS_SR + [("dL_dsh", "shs")]
    if kind == "sh1":
        shs = torch.tensor((rng.normal(size=(P, 4, 3)) * 0.5).astype(np.float32))
        return 3, dict(shs=shs, sh_degree=1), GRADS_SR + [("dL_dsh", "shs")]
    if kind == "cov":
        A = rng.normal(size=(P, 3, 3)) * 0.12
        S = A @ A.transpose(0, 2, 1) + 1e-4 * np.eye(3)
        cov = torch.tensor(np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1)
                           .astype(np.float32))
        colors = torch.tensor(rng.normal(size=(P, 3)).astype(np.float32))
        return 3, dict(colors=colors, cov3D=cov), [("dL_dmeans3D", "means3D"), ("dL_dcov3D", "cov3D"),
                                                     ("dL_dopacity", "opacities"), ("dL_dcolors", "colors"),
                                                     ("dL_dmean2D", "means2D")]
    raise ValueError(kind)


def test_sh_basis_matches_reference_fixture():
    """ref_splat's Legendre-recurrence SH against eval_sh outputs of the reference itself
    (tests/golden/reference_fixtures.json, generated by importing utils/sh_utils.py)."""
    fx = json.load(open(os.path.join(HERE, "golden", "reference_fixtures.json")))
    n = 0
    for case in fx["sh"]:
        deg = case["deg"]
        sh = torch.tensor(case["sh"], dtype=F64)          # (N, 3, (deg+1)^2) as eval_sh takes it
        dirs = torch.tensor(case["dirs"], dtype=F64)
        want = torch.tensor(case["result"], dtype=F64)
        Y = ref_splat.real_sh_basis(deg, dirs)
        got = torch.einsum("nk,nck->nc", Y, sh[..., : Y.shape[1]])
        assert (got - want).abs().max() < 2e-6, deg
        n += 1
    assert n >= 4


@pytest.mark.parametrize("kind", ["precomp", "sh", "sh1", "cov"])
def test_oracle_matches_independent_splat(orc, kind):
    sc = pin_scene(seed={"precomp": 1, "sh": 2, "sh1": 3, "cov": 4}[kind])
    C, inp, names = _case(sc, kind)
    bg = sc["rng"].normal(size=C).astype(np.float32)
    dL = sc["rng"].normal(size=(C, sc["H"], sc["W"])).astype(np.float32)
    r, ref = _ref(sc, C, bg, dL, **inp)
    fw, g = _oracle(orc, sc, C, bg, dL, **inp)
    assert (~r["clamp_inside"].numpy()).any(1).sum() >= 6          # the clamp really is active
    assert (r["radii"].numpy()[: 8] > 0).sum() >= 4                  # ... on Gaussians that are rendered
    _check_forward(fw, r)
    _check_grads(g, ref, names)


def test_scale_modifier_gradient_convention(orc):
    """The reference returns dL/d(mod * scale), not dL/dscale (backward.cu:289-318 never multiplies by
    `mod`): with scale_modifier = 1.7 the oracle's dL_dscales is autograd's divided by 1.7."""
    sc = pin_scene(seed=7)
    C, inp, _ = _case(sc, "precomp")
    bg = np.zeros(C, np.float32)
    dL = sc["rng"].normal(size=(C, sc["H"], sc["W"])).astype(np.float32)
    _, ref = _ref(sc, C, bg, dL, mod=1.7, **inp)
    _, g = _oracle(orc, sc, C, bg, dL, mod=1.7, **inp)
    assert _rel(g["dL_dscales"], ref["scales"] / 1.7) < GRAD_TOL
    assert _rel(g["dL_drotations"], ref["rotations"]) < GRAD_TOL
    assert _rel(g["dL_dmeans3D"], ref["means3D"]) < GRAD_TOL


def test_hand_kat_anisotropic_rotated_offaxis(orc):
    """Appendix-B style known answer that is NOT invariant under transposition mistakes: one Gaussian with
    scales (0.30, 0.05, 0.02) and quaternion (1,1,1,1)/2 -- the cyclic permutation of the axes, a rotation
    whose matrix differs from its transpose -- off axis at p = (0.5, -0.3, 2), camera at the origin.
    Every expected number below is worked out by hand from the textbook formulas."""
    W, H, fx = 1296, 968, 1170.0
    from sgs_hip.camera import pinhole
    cam = pinhole(W, H, fx)
    # q = q_x(90) * q_z(90)  (apply z rotation first): (w,x,y,z) = (0.5, 0.5, 0.5, 0.5)  [cyclic permutation]
    q = np.array([[0.5, 0.5, 0.5, 0.5]], np.float32)
    s = np.array([[0.30, 0.05, 0.02]], np.float32)
    p = np.array([[0.5, -0.3, 2.0]], np.float32)
    # R for q = (1,1,1,1)/2 is the cyclic permutation e1 -> e2 -> e3 -> e1, so world variances are
    # (var_x, var_y, var_z) = (s3^2, s1^2, s2^2) = (0.0004, 0.09, 0.0025)
    pre = orc.preprocess(p, np.array([[0.5]], np.float32), cam.world_view_transform.numpy(),
                         cam.full_proj_transform.numpy(), cam.camera_center.numpy(), W, H, cam.tanfovx,
                         cam.tanfovy, scales=s, rotations=q, colors_precomp=np.ones((1, 3), np.float32))
    assert np.allclose(pre["cov3D"][0], [0.0004, 0, 0, 0.09, 0, 0.0025], atol=1e-7)
    # J = [[fx/z, 0, -fx x/z^2], [0, fy/z, -fy y/z^2]] = [[585, 0, -146.25], [0, 585, 87.75]]
    # Sigma2 = J diag(0.0004, 0.09, 0.0025) J^T + 0.3 I
    a = 585.0 ** 2 * 0.0004 + 146.25 ** 2 * 0.0025 + 0.3          # 190.662...
    b = -146.25 * 87.75 * 0.0025                                   # -32.0836...
    c = 585.0 ** 2 * 0.09 + 87.75 ** 2 * 0.0025 + 0.3              # 30820.0...
    det = a * c - b * b
    want = np.array([c / det, -b / det, a / det])
    assert np.allclose(pre["conic_opacity"][0, :3], want, rtol=2e-5)
    # pixel: ndc = (x/z) / tanfov -> ((ndc + 1) S - 1) / 2
    assert abs(pre["means2D"][0, 0] - (((0.25 * fx / (W / 2)) + 1) * W - 1) / 2) < 1e-2
    assert abs(pre["means2D"][0, 1] - (((-0.15 * fx / (H / 2)) + 1) * H - 1) / 2) < 1e-2
    lam = 0.5 * (a + c) + math.sqrt(max(0.1, (0.5 * (a + c)) ** 2 - det))
    assert pre["radii"][0] == math.ceil(3 * math.sqrt(lam))       # 527: the long axis is vertical on screen


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["precomp", "sh", "sh1", "cov"])
def test_hip_matches_independent_splat(kind):
    """The same pin for the HIP path: forward per-Gaussian state and the full backward through
    sgs_rasterize_forward / sgs_rasterize_backward (C-ABI) against float64 autograd."""
    from sgs_hip import raster
    sc = pin_scene(seed={"precomp": 1, "sh": 2, "sh1": 3, "cov": 4}[kind])
    C, inp, names = _case(sc, kind)
    bg = sc["rng"].normal(size=C).astype(np.float32)
    dL = sc["rng"].normal(size=(C, sc["H"], sc["W"])).astype(np.float32)
    r, ref = _ref(sc, C, bg, dL, **inp)
    dev = "cuda:0"
    cam = sc["cam"].to(dev)
    e = torch.Tensor([])
    t = lambda k: inp[k].to(dev) if k in inp else e
    W, H, P = sc["W"], sc["H"], sc["means3D"].shape[0]
    m3, op = sc["means3D"].to(dev), sc["opacities"].to(dev)
    scales = e if "cov3D" in inp else sc["scales"].to(dev)
    rots = e if "cov3D" in inp else sc["rotations"].to(dev)
    deg = inp.get("sh_degree", 0)
    bgt = torch.tensor(bg, device=dev)
    n, color, radii, geom, binn, img, _ = raster.rasterize_forward(
        bgt, m3, t("colors"), op, scales, rots, 1.0, t("cov3D"), cam.world_view_transform,
        cam.full_proj_transform, cam.tanfovx, cam.tanfovy, H, W, t("shs"), deg, cam.camera_center, False,
        False, C, False)
    gv = raster.geometry_views(geom, P)
    fw = dict(radii=radii.cpu().numpy(), depths=gv["depths"].cpu().numpy(), means2D=gv["means2D"].cpu().numpy(),
              conic_opacity=gv["conic_opacity"].cpu().numpy(), out=color.cpu().numpy())
    _check_forward(fw, r)
    grads = raster.rasterize_backward(bgt, m3, radii, t("colors"), scales, rots, 1.0, t("cov3D"),
                                      cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy,
                                      torch.tensor(dL, device=dev), t("shs"), deg, cam.camera_center, geom, n,
                                      binn, img, False)
    keys = ["dL_dmean2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales",
            "dL_drotations"]
    g = {k: v.cpu().numpy() for k, v in zip(keys, grads)}
    _check_grads(g, ref, names)

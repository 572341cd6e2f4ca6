"""CPU tests that pin the oracle: the known-answer tests of SURVEY.md Appendix B and the
fixtures generated from the reference's own Python helpers (tests/golden/)."""
import json
import math
import os

import numpy as np
import pytest
import torch

from sgs_hip.camera import make_camera, pinhole, focal2fov

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _one_gaussian(orc, p, s, opacity, W=1296, H=968, fx=1170.0, C=1):
    cam = pinhole(W, H, fx)
    means = np.array([p], np.float32)
    fw = orc.forward(means, np.array([[opacity]], np.float32), cam.world_view_transform.numpy(),
                     cam.full_proj_transform.numpy(), cam.camera_center.numpy(), W, H,
                     cam.tanfovx, cam.tanfovy, np.zeros(C, np.float32), C,
                     scales=np.array([[s, s, s]], np.float32),
                     rotations=np.array([[1, 0, 0, 0]], np.float32),
                     colors_precomp=np.ones((1, C), np.float32))
    return fw, cam


def test_camera_convention_kat():
    cam = make_camera(np.eye(3), np.array([0, 0, 3.0]), focal2fov(500, 640), focal2fov(500, 480), 640, 480)
    wvt = cam.world_view_transform.numpy()
    assert np.allclose(wvt, np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 3, 1]]), atol=1e-6)
    fp = cam.full_proj_transform.numpy()
    exp = np.array([[1.5625, 0, 0, 0], [0, 2.083333, 0, 0], [0, 0, 1.0001, 1], [0, 0, 2.9903, 3]])
    assert np.allclose(fp, exp, atol=2e-4)
    assert np.allclose(cam.camera_center.numpy(), [0, 0, -3], atol=1e-6)


def test_cameras_match_reference_fixtures():
    fx = json.load(open(os.path.join(GOLD, "reference_fixtures.json")))
    for c in fx["cameras"]:
        cam = make_camera(np.array(c["R"]), np.array(c["T"]), c["FoVx"], c["FoVy"], c["W"], c["H"])
        assert np.array_equal(cam.world_view_transform.numpy().astype(np.float64), np.array(c["world_view_transform"]))
        assert np.array_equal(cam.full_proj_transform.numpy().astype(np.float64), np.array(c["full_proj_transform"]))
        assert np.array_equal(cam.camera_center.numpy().astype(np.float64), np.array(c["camera_center"]))


def test_sh_dense_directions_match_reference_eval_sh(orc):
    """kernel and oracle evaluate the same GENERATED monomial table (tools/gen_sh_table.py), so their bit-exact agreement
    cannot catch a generator bug.  1 024 known answers of the reference's own eval_sh (utils/sh_utils.py, float64; fixture
    tests/golden/sh_dense.npz, generator beside it): the oracle's SH -> RGB within a few fp32 ulps of the value's scale
    (sum_n |Y_n sh_n| + 0.5), the clamp flags equal wherever the unclamped value is not within that bound of zero."""
    z = np.load(os.path.join(GOLD, "sh_dense.npz"))
    cam = pinhole(64, 64, 3.0)   # (a very wide view: the directions cover most of the front hemisphere)
    worst = 0.0
    for deg in range(4):
        sh, pos, ref = z[f"sh{deg}"], z[f"pos{deg}"], z[f"res{deg}"]
        n = sh.shape[0]
        shs = np.ascontiguousarray(sh.transpose(0, 2, 1))   # rasteriser layout (P, M, 3)
        pre = orc.preprocess(pos, np.full((n, 1), 0.5, np.float32), cam.world_view_transform.numpy(),
                             cam.full_proj_transform.numpy(), np.zeros(3, np.float32), 64, 64, cam.tanfovx, cam.tanfovy,
                             scales=np.full((n, 3), 0.05, np.float32), rotations=np.tile(np.array([[1, 0, 0, 0]], np.float32), (n, 1)),
                             shs=shs, sh_degree=deg)
        vis = pre["radii"] > 0
        assert vis.sum() > n // 2
        nb = (deg + 1) ** 2
        scale = np.abs(sh[:, :, :nb]).sum(axis=2) + 0.5       # |Y_n| <= ~1.5: the terms' magnitude
        want = ref + 0.5
        got = pre["rgb"]
        tol = 6.0 * 2.0 ** -23 * scale
        err = np.abs(got.astype(np.float64) - np.maximum(want, 0.0))
        assert (err[vis] <= tol[vis]).all(), (deg, float((err[vis] / scale[vis]).max()))
        worst = max(worst, float((err[vis] / scale[vis]).max()))
        sure = np.abs(want) > tol
        assert np.array_equal(pre["clamped"].astype(bool)[vis & sure.all(axis=1)], (want < 0)[vis & sure.all(axis=1)])
    print(f"dense SH known answers: worst |oracle - reference| = {worst:.2e} of the terms' magnitude")


def test_cov3d_matches_reference_python_route(orc):
    """One more formula of the path pinned by the reference itself: the 3-D covariance.  The reference has two routes to it -- the kernel's
    computeCov3D (CR/cuda_rasterizer/forward.cu:115-150) and `pipe.compute_cov3D_python` = GaussianModel.get_covariance
    (model/gaussian_model.py:34-38: strip_symmetric(L L^T), L = R(q) diag(modifier * s), utils/general_utils.py:66-115), which must agree.
    512 known answers of the Python route (fixture tests/golden/cov3d_reference.npz, generator beside it: the reference's helpers imported,
    scales over four decades, three scale modifiers): the oracle's cov3D within 16 fp32 ulps of the entry's scale |R|^2 (modifier s)^2."""
    z = np.load(os.path.join(GOLD, "cov3d_reference.npz"))
    scales, rots = z["scales"], z["rotations"]
    n = scales.shape[0]
    cam = pinhole(64, 64, 30.0)
    pos = np.tile(np.array([[0.0, 0.0, 3.0]], np.float32), (n, 1))   # on the axis: nothing is culled
    worst = 0.0
    for i, m in enumerate(z["modifiers"]):
        pre = orc.preprocess(pos, np.full((n, 1), 0.5, np.float32), cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(),
                             np.zeros(3, np.float32), 64, 64, cam.tanfovx, cam.tanfovy, scales=scales, rotations=rots,
                             scale_modifier=float(m), colors_precomp=np.ones((n, 3), np.float32))
        vis = pre["radii"] > 0
        assert vis.sum() > n * 3 // 4   # (the smallest Gaussians project to a zero radius: their record is not written)
        want = z[f"cov3D_{i}"].astype(np.float64)
        mag = (float(m) * scales.astype(np.float64).max(axis=1, keepdims=True)) ** 2   # every entry is a sum of products of (m s_k) R_ik R_jk, |R| <= 1
        err = np.abs(pre["cov3D"].astype(np.float64) - want) / mag
        assert (err[vis] <= 16.0 * 2.0 ** -23).all(), (float(m), float(err[vis].max()))   # (measured: 7.8 ulp -- two fp32 routes with different operation orders)
        worst = max(worst, float(err[vis].max()))
    print(f"cov3D known answers: worst |oracle - reference| = {worst:.2e} of (modifier * largest scale)^2")


def test_sh_matches_reference_eval_sh(orc):
    """oracle SH->RGB (before +0.5/clamp) == utils/sh_utils.py eval_sh (fixture)."""
    fx = json.load(open(os.path.join(GOLD, "reference_fixtures.json")))
    for case in fx["sh"]:
        deg = case["deg"]
        sh = np.array(case["sh"], np.float32)        # (n,3,16) eval_sh layout
        dirs = np.array(case["dirs"], np.float32)
        ref = np.array(case["result"], np.float32)
        n = sh.shape[0]
        # rasteriser layout is (P, M, 3); place the camera at the origin and the point at `dir`
        shs = np.ascontiguousarray(sh.transpose(0, 2, 1))
        cam = pinhole(64, 64, 50.0)
        means = dirs * 1.0
        means[:, 2] = np.abs(means[:, 2]) + 1.0      # in front of the camera
        d = means / np.linalg.norm(means, axis=1, keepdims=True)
        pre = orc.preprocess(means, np.full((n, 1), 0.5, np.float32), cam.world_view_transform.numpy(),
                             cam.full_proj_transform.numpy(), np.zeros(3, np.float32), 64, 64,
                             cam.tanfovx, cam.tanfovy, scales=np.full((n, 3), 0.5, np.float32),
                             rotations=np.tile(np.array([[1, 0, 0, 0]], np.float32), (n, 1)),
                             shs=shs, sh_degree=deg)
        from sys import path as _p  # noqa: F401
        # recompute the reference polynomial for the actual directions with the fixture's coeffs
        # by linearity in sh: compare against the fixture when directions coincide
        vis = pre["radii"] > 0
        assert vis.any()
        # direct check at the fixture's own directions: build points exactly along them
        means2 = dirs.copy()
        flip = means2[:, 2] < 0.25
        means2[flip] *= -1.0                          # eval_sh is evaluated at -d for flipped ones
        keep = means2[:, 2] > 0.25
        pre2 = orc.preprocess(means2, np.full((n, 1), 0.5, np.float32), cam.world_view_transform.numpy(),
                              cam.full_proj_transform.numpy(), np.zeros(3, np.float32), 64, 64,
                              cam.tanfovx, cam.tanfovy, scales=np.full((n, 3), 0.5, np.float32),
                              rotations=np.tile(np.array([[1, 0, 0, 0]], np.float32), (n, 1)),
                              shs=shs, sh_degree=deg)
        sel = keep & ~flip & (pre2["radii"] > 0)
        assert sel.sum() >= 3
        got = pre2["rgb"][sel]
        want = np.maximum(ref[sel] + 0.5, 0.0)
        assert np.allclose(got, want, rtol=0, atol=2e-6)
        assert np.array_equal(pre2["clamped"][sel].astype(bool), (ref[sel] + 0.5) < 0)


def test_projection_on_axis_point(orc):
    fw, cam = _one_gaussian(orc, (0, 0, 2.0), 0.05, 0.5)
    assert abs(cam.tanfovx - 1296 / (2 * 1170.0)) < 1e-6
    assert np.allclose(fw["means2D"][0], [647.5, 483.5])
    assert fw["depths"][0] == np.float32(2.0)


def test_radius_rect_key_kat(orc):
    fw, _ = _one_gaussian(orc, (0, 0, 2.0), 0.05, 0.5)
    # Sigma2 diag = (1170*0.05/2)^2 + 0.3 = 855.8625 ; lambda1 = mid + sqrt(0.1) ; radius 88
    co = fw["conic_opacity"][0]
    assert abs(1.0 / co[0] - 855.8625) < 1e-2 and abs(co[1]) < 1e-9
    assert fw["radii"][0] == 88
    assert fw["tiles_touched"][0] == 144 and fw["num_rendered"] == 144
    keys = fw["keys_unsorted"]
    tiles = (keys >> np.uint64(32)).astype(np.int64)
    assert tiles[0] == 24 * 81 + 34
    assert int(keys[0]) == ((24 * 81 + 34) << 32) | 0x40000000      # depth bits of 2.0f
    tx, ty = tiles % 81, tiles // 81
    assert tx.min() == 34 and tx.max() == 45 and ty.min() == 24 and ty.max() == 35
    # emission order: row-major y then x
    assert np.array_equal(tiles, np.array([y * 81 + x for y in range(24, 36) for x in range(34, 46)]))


def test_centre_pixel_alpha_kat(orc):
    fw, _ = _one_gaussian(orc, (0, 0, 2.0), 0.05, 0.5)
    # pixel (647,483): d = (0.5,0.5), power = -0.000292103, alpha = 0.49985397
    out = fw["out"][0, 483, 647]
    assert abs(out - 0.499853970) < 2e-6          # colour 1 * alpha * T(=1)
    assert abs(fw["final_T"][483, 647] - (1 - 0.499853970)) < 2e-6
    assert fw["n_contrib"][483, 647] == 1


def test_higher_msb_kat(orc):
    for n, want in zip([1, 2, 3, 4, 255, 256, 257, 4941, 65535, 65536], [1, 2, 2, 3, 8, 9, 9, 13, 16, 17]):
        assert orc.higher_msb(n) == want


def test_morton_spread_kat(orc):
    assert orc.prep_morton(1) == 1 and orc.prep_morton(2) == 8 and orc.prep_morton(1023) == 0x09249249


def test_knn_unit_cube(orc):
    pts = np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)], np.float32)
    assert np.array_equal(orc.dist2(pts), np.ones(8, np.float32))


def test_knn_duplicates_and_small_p(orc):
    pts = np.array([[0, 0, 0], [0, 0, 0], [1, 0, 0], [0, 2, 0], [5, 5, 5]], np.float32)
    d = orc.dist2(pts)
    assert d[0] == np.float32((0 + 1 + 4) / 3.0)         # duplicate at distance 0 counts
    d3 = orc.dist2(pts[:3])                                # P < 4: a FLT_MAX slot stays in the sum
    fmax = np.finfo(np.float32).max
    assert d3[0] == (np.float32(0) + np.float32(1) + fmax) / np.float32(3.0)
    assert (d3 > 1e37).all()


def test_empty_tile_and_background(orc):
    fw, _ = _one_gaussian(orc, (0, 0, 2.0), 0.05, 0.5, C=2)
    # far corner tile is untouched: out = bg (here 0), T = 1, n = 0
    assert fw["out"][:, 0, 0].tolist() == [0.0, 0.0]
    assert fw["final_T"][0, 0] == 1.0 and fw["n_contrib"][0, 0] == 0
    assert fw["ranges"][0].tolist() == [0, 0]
    cam = pinhole(64, 48, 60.0)
    bg = np.array([0.25, -1.5, 3.0], np.float32)
    fw2 = orc.forward(np.array([[0, 0, -5.0]], np.float32), np.array([[0.9]], np.float32),
                      cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(),
                      cam.camera_center.numpy(), 64, 48, cam.tanfovx, cam.tanfovy, bg, 3,
                      scales=np.ones((1, 3), np.float32), rotations=np.array([[1, 0, 0, 0]], np.float32),
                      colors_precomp=np.ones((1, 3), np.float32))
    assert fw2["num_rendered"] == 0 and (fw2["radii"] == 0).all()   # behind the camera: culled
    assert np.array_equal(fw2["out"], np.broadcast_to(bg[:, None, None], (3, 48, 64)))


def test_empty_scene_returns_zeros(orc):
    cam = pinhole(32, 32, 30.0)
    fw = orc.forward(np.zeros((0, 3), np.float32), np.zeros((0, 1), np.float32),
                     cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(),
                     cam.camera_center.numpy(), 32, 32, cam.tanfovx, cam.tanfovy,
                     np.ones(3, np.float32), 3, colors_precomp=np.zeros((0, 3), np.float32),
                     scales=np.zeros((0, 3), np.float32), rotations=np.zeros((0, 4), np.float32))
    assert fw["num_rendered"] == 0 and not fw["out"].any()


def test_expf_contract_accuracy(orc):
    x = np.concatenate([np.linspace(-30, 0, 20001), -np.logspace(-8, 1.9, 500)]).astype(np.float32)
    got = orc.expf(x).astype(np.float64)
    want = np.exp(x.astype(np.float64))
    rel = np.abs(got - want) / want
    assert rel.max() < 4e-7                  # <= ~6 ulp, far inside the 1e-4 budget
    assert orc.expf(np.float32(0.0)) == np.float32(1.0)
    assert orc.expf(np.float32(-1000.0)) < 1e-37   # clamped at exp(-87)


def test_sort_is_stable_on_key_bits(orc):
    import ctypes as C
    rng = np.random.default_rng(3)
    L = 5000
    tiles = rng.integers(0, 7, L).astype(np.uint64)
    depth = rng.integers(0, 5, L).astype(np.uint64)       # many ties
    junk = rng.integers(0, 4, L).astype(np.uint64) << np.uint64(40)   # bits above end_bit must be ignored
    keys = (tiles << np.uint64(32)) | depth | junk
    vals = np.arange(L, dtype=np.uint32)
    ko, vo = np.zeros(L, np.uint64), np.zeros(L, np.uint32)
    orc.lib().orc_sort_pairs(C.c_size_t(L), orc._p(keys), orc._p(vals), orc._p(ko), orc._p(vo), C.c_int(35))
    masked = keys & np.uint64((1 << 35) - 1)
    order = np.argsort(masked, kind="stable")
    assert np.array_equal(vo, vals[order]) and np.array_equal(ko, keys[order])


def test_oracle_blend_matches_float64_composite(orc):
    """Independent check of the blend restatement: a dense float64 numpy composite of the
    oracle's own sorted lists (same skip/stop rules) must agree to 1e-5."""
    from helpers import small_scene, oracle_forward
    scene, cam = small_scene(P=600, C=5, W=64, H=48, fx=60.0, seed=2)
    fw = oracle_forward(orc, scene, cam)
    W, H = cam.image_width, cam.image_height
    gx = (W + 15) // 16
    out = np.zeros((5, H, W))
    for y in range(H):
        for x in range(W):
            t = (y // 16) * gx + x // 16
            r0, r1 = fw["ranges"][t]
            T = 1.0
            for e in range(r0, r1):
                g = fw["point_list"][e]
                dx, dy = fw["means2D"][g].astype(np.float64) - (x, y)
                a, b, c, o = fw["conic_opacity"][g].astype(np.float64)
                power = -0.5 * (a * dx * dx + c * dy * dy) - b * dx * dy
                if power > 0:
                    continue
                alpha = min(0.99, o * math.exp(power))
                if alpha < 1 / 255:
                    continue
                if T * (1 - alpha) < 1e-4:
                    break
                out[:, y, x] += fw["features"][g].astype(np.float64) * alpha * T
                T *= 1 - alpha
            # bg = 0
    assert np.abs(out - fw["out"]).max() < 1e-5


def test_oracle_backward_matches_autograd(orc):
    """The runtime-C backward restatement against torch autograd of a differentiable float64
    splat built from the oracle's own lists."""
    from helpers import small_scene, oracle_forward
    scene, cam = small_scene(P=160, C=4, W=32, H=24, fx=30.0, seed=5)
    scene = scene._replace(opacities=scene.opacities.clamp(max=0.95))   # stay below the 0.99 clamp
    bg = np.array([0.3, -0.2, 0.1, 0.7], np.float32)
    fw = oracle_forward(orc, scene, cam, bg=bg)
    W, H = cam.image_width, cam.image_height
    rng = np.random.default_rng(0)
    dL = rng.normal(size=(4, H, W)).astype(np.float32)
    g = orc.blend_backward(fw, bg, dL, W, H)

    m2d = torch.tensor(fw["means2D"], dtype=torch.float64, requires_grad=True)
    conic = torch.tensor(fw["conic_opacity"][:, :3], dtype=torch.float64, requires_grad=True)
    opac = torch.tensor(fw["conic_opacity"][:, 3], dtype=torch.float64, requires_grad=True)
    feats = torch.tensor(fw["features"], dtype=torch.float64, requires_grad=True)
    gx = (W + 15) // 16
    loss = torch.zeros((), dtype=torch.float64)
    bgt = torch.tensor(bg, dtype=torch.float64)
    for y in range(H):
        for x in range(W):
            t = (y // 16) * gx + x // 16
            r0, _ = fw["ranges"][t]
            n = int(fw["n_contrib"][y, x])
            T = torch.ones((), dtype=torch.float64)
            col = torch.zeros(4, dtype=torch.float64)
            for e in range(r0, r0 + n):
                gi = int(fw["point_list"][e])
                d = m2d[gi] - torch.tensor([x, y], dtype=torch.float64)
                power = -0.5 * (conic[gi, 0] * d[0] * d[0] + conic[gi, 2] * d[1] * d[1]) - conic[gi, 1] * d[0] * d[1]
                if power.item() > 0:
                    continue
                alpha_raw = opac[gi] * torch.exp(power)
                alpha = torch.clamp(alpha_raw, max=0.99)
                if alpha.item() < 1 / 255:
                    continue
                col = col + feats[gi] * alpha * T
                T = T * (1 - alpha)
            col = col + T * bgt
            loss = loss + (col * torch.tensor(dL[:, y, x], dtype=torch.float64)).sum()
    loss.backward()
    # the reference's dL/dmean2D is w.r.t. NDC-scaled coordinates: factor 0.5*W / 0.5*H
    # and has NO gradient mask for the 0.99 clamp; none of the alphas here reach 0.99
    assert (fw["conic_opacity"][:, 3] < 0.98).all()
    assert fw["n_contrib"].max() > 5

    def close(a, b, tol=2e-4):
        scale = max(np.abs(b).max(), 1e-12)
        return np.abs(a - b).max() / scale < tol

    assert close(g["dL_dcolors"], feats.grad.numpy())
    assert close(g["dL_dopacity"][:, 0], opac.grad.numpy())
    assert close(g["dL_dmean2D"][:, 0], m2d.grad.numpy()[:, 0] * 0.5 * W)
    assert close(g["dL_dmean2D"][:, 1], m2d.grad.numpy()[:, 1] * 0.5 * H)
    assert close(g["dL_dconic"][:, 0], conic.grad.numpy()[:, 0])
    # reference convention: the off-diagonal slot carries HALF the derivative w.r.t. b (the
    # symmetric entry appears twice; computeCov2D's backward re-doubles it, backward.cu:209)
    assert close(g["dL_dconic"][:, 1], 0.5 * conic.grad.numpy()[:, 1])
    assert close(g["dL_dconic"][:, 3], conic.grad.numpy()[:, 2])

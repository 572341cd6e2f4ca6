"""Every BASELINE.json configuration on the HIP path (cfg3's own full-size tests live in test_fullsize_gpu.py):

  cfg1  10k Gaussians, 256x256, C = 3: the whole frame against the oracle (and the pure-PyTorch CPU splat);
  cfg2  500k Gaussians, RGB-D, 968x1296 and 484x648: every integer output of the full frame against the oracle,
        colour + median depth on sampled tile rows, bit for bit;
  cfg4  5M Gaussians x 768 channels (P * C > 2^31: the reference's int index overflows, CR/forward.cu:356), 840x1297
        (a width that is not a multiple of 16): integers in full, sampled tile rows against the oracle;
  cfg5  Gaussian sharding as a depth-ordered composite of HIP (A, T) partials against the single render on one GPU: a
        two-slab composite at P = 2M, and cfg5 at its stated 50M Gaussians x 256 channels as eight depth slabs through the
        HIP composite kernel (test_cfg5_full_size_on_one_gpu);
  cfg3  the default (six-product, f32-equivalent) arithmetic DIRECTLY against the oracle and against the exact (float64)
        composite: all 512 channels on four tile rows, element-wise.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
E = torch.Tensor([])


def _oracle_front(orc, scene, cam, W, H):
    pre = orc.preprocess(scene.means3D.numpy(), scene.opacities.numpy(), cam.world_view_transform.numpy(),
                         cam.full_proj_transform.numpy(), cam.camera_center.numpy(), W, H, cam.tanfovx, cam.tanfovy,
                         scales=scene.scales.numpy(), rotations=scene.rotations.numpy(),
                         colors_precomp=np.zeros((1, 1), np.float32))
    return pre, orc.binning(pre, W, H)


def _forward(s, c, feats, bg, W, H, want_depth=False, pool=None):
    from sgs_hip import raster
    return raster.rasterize_forward(bg, s.means3D, feats, s.opacities, s.scales, s.rotations, 1.0, E,
                                    c.world_view_transform, c.full_proj_transform, c.tanfovx, c.tanfovy, H, W, E, 0,
                                    c.camera_center, False, False, feats.shape[1], want_depth, pool=pool)


def _check_integers(raster, out, pre, binn, P, W, H):
    n, color, radii, geom, bbuf, img, depth = out
    assert n == binn["num_rendered"]
    assert np.array_equal(radii.cpu().numpy(), pre["radii"])
    b = raster.binning_views(bbuf, n, geom, P, img, W, H)
    assert np.array_equal(b["point_list"].cpu().numpy().view(np.uint32), binn["point_list"])
    assert np.array_equal(b["keys_sorted"].cpu().numpy().view(np.uint64), binn["keys_sorted"])
    iv = raster.image_views(img, W, H)
    assert np.array_equal(iv["ranges"].cpu().numpy().view(np.uint32), binn["ranges"])
    return iv


def test_cfg1_full_frame(orc):
    from oracle import torch_splat
    from sgs_hip import raster
    from sgs_hip.synthetic import make_config
    scene, cam = make_config("cfg1")
    W = H = 256
    pre, binn = _oracle_front(orc, scene, cam, W, H)
    ob = orc.blend_forward(pre, binn, scene.features.numpy(), scene.bg.numpy(), W, H, want_depth=True)
    s, c = scene.to(DEV), cam.to(DEV)
    out = _forward(s, c, s.features, s.bg, W, H, want_depth=True)
    iv = _check_integers(raster, out, pre, binn, 10_000, W, H)
    assert np.array_equal(out[1].cpu().numpy().view(np.uint32), ob["out"].view(np.uint32))      # RGB, bit for bit
    assert np.array_equal(out[6].cpu().numpy().view(np.uint32), ob["depth"].view(np.uint32))    # median depth
    assert np.array_equal(iv["n_contrib"].cpu().numpy().view(np.uint32), ob["n_contrib"])
    chn = _forward(s, c, s.features, s.bg, W, H)                                                  # N-channel package, C = 3
    assert torch.equal(chn[1], out[1])
    ts = torch_splat.render(scene, cam, W, H)                                                     # config 1's CPU renderer
    err = (ts["out"] - out[1].cpu()).abs()
    assert float(torch.quantile(err.flatten()[::7], 0.999)) < 2e-6


@pytest.mark.parametrize("name", ["cfg2", "cfg2_half"])
def test_cfg2_rgbd(orc, name):
    from sgs_hip import raster
    from sgs_hip.synthetic import CONFIGS, make_config
    P, C, W, H, fx = CONFIGS[name]
    scene, cam = make_config(name)
    pre, binn = _oracle_front(orc, scene, cam, W, H)
    s, c = scene.to(DEV), cam.to(DEV)
    out = _forward(s, c, s.features, s.bg, W, H, want_depth=True)
    iv = _check_integers(raster, out, pre, binn, P, W, H)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    for row in (0, gy // 2, gy - 1):           # top, middle and the (ragged) bottom tile row
        ob = orc.blend_forward(pre, binn, scene.features.numpy(), scene.bg.numpy(), W, H, want_depth=True,
                               tile_lo=row * gx, tile_hi=(row + 1) * gx)
        rows = slice(row * 16, min(H, row * 16 + 16))
        assert np.array_equal(out[1][:, rows].cpu().numpy().view(np.uint32), ob["out"][:, rows].view(np.uint32))
        assert np.array_equal(out[6][:, rows].cpu().numpy().view(np.uint32), ob["depth"][:, rows].view(np.uint32))
        assert np.array_equal(iv["n_contrib"][rows].cpu().numpy().view(np.uint32), ob["n_contrib"][rows])


def test_cfg4_5m_gaussians_768_channels_64bit_indexing(orc):
    from sgs_hip import raster
    from sgs_hip.synthetic import CONFIGS, make_config
    P, C, W, H, fx = CONFIGS["cfg4"]
    assert P * C > 2 ** 31 and W % 16 != 0
    scene, cam = make_config("cfg4", features=False)               # geometry from the frozen generator
    g = torch.Generator(device=DEV).manual_seed(4)
    feats = torch.randn(P, C, device=DEV, generator=g)             # the 15.4 GB feature table, made on the device
    feats /= feats.norm(dim=1, keepdim=True)
    bg = torch.linspace(-1.0, 1.0, C, device=DEV)
    pre, binn = _oracle_front(orc, scene, cam, W, H)
    s, c = scene._replace(features=torch.empty(0, C)).to(DEV), cam.to(DEV)
    pool = raster.ScratchPool()
    out = _forward(s, c, feats, bg, W, H, pool=pool)
    iv = _check_integers(raster, out, pre, binn, P, W, H)
    default = out[1]
    raster.set_blend_exact(True)
    try:
        exact = _forward(s, c, feats, bg, W, H, pool=pool)[1]
    finally:
        raster.set_blend_exact(False)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    assert gx * 16 > W                                              # the last tile column is 1 pixel wide
    feats_host = feats.cpu().numpy()
    high = 0
    for row in (1, gy // 2):
        lo, hi = row * gx, (row + 1) * gx
        ob = orc.blend_forward(pre, binn, feats_host, bg.cpu().numpy(), W, H, tile_lo=lo, tile_hi=hi)
        rows = slice(row * 16, row * 16 + 16)
        want = ob["out"][:, rows]
        assert np.array_equal(exact[:, rows].cpu().numpy().view(np.uint32), want.view(np.uint32))
        assert np.abs(default[:, rows].cpu().numpy() - want).max() <= 1e-4 * np.abs(want).max()
        assert np.array_equal(iv["n_contrib"][rows].cpu().numpy().view(np.uint32), ob["n_contrib"][rows])
        r = binn["ranges"][lo:hi]
        ids = np.concatenate([binn["point_list"][a:b] for a, b in r])
        high += int((ids.astype(np.int64) * C >= 2 ** 31).sum())
    assert high > 1000       # the sampled rows really do gather rows beyond the 32-bit element index


def test_cfg5_gaussian_sharding_two_depth_slabs(orc):
    """BASELINE config 5's data flow on one device: two depth slabs -> HIP (A, T) partials -> "over" composite,
    against the single render of all Gaussians."""
    from sgs_hip import raster, dist as sdist
    from sgs_hip.synthetic import CONFIGS, make_scene
    from sgs_hip.camera import pinhole
    _, C, W, H, fx = CONFIGS["cfg5"]
    P = 2_000_000
    scene = make_scene(P, C, W, H, fx, seed=5)
    cam = pinhole(W, H, fx)
    s, c = scene.to(DEV), cam.to(DEV)
    bg = torch.linspace(0.0, 1.0, C, device=DEV)
    n, whole, _, _, _, img, _ = _forward(s, c, s.features, bg, W, H)
    T_whole = raster.image_views(img, W, H)["final_T"].clone()
    depth = s.means3D[:, 2]                                        # camera at the origin looking down +z
    cut = float(depth.median())
    partials = []
    for mask in (depth < cut, depth >= cut):                       # front slab, back slab
        A, T, radii = raster.render_partial(s.means3D[mask], s.features[mask], s.opacities[mask], s.scales[mask],
                                            s.rotations[mask], c.world_view_transform, c.full_proj_transform,
                                            c.tanfovx, c.tanfovy, H, W, c.camera_center)
        assert int((radii > 0).sum()) > 100_000
        partials.append((A, T))
    comp, t_total = sdist.composite_over(partials, bg)
    scale = float(whole.abs().max())
    # The only systematic difference from the single render is the reference's stop rule: a pixel is DONE when the
    # next entry would push T below 1e-4 -- and that entry (alpha up to 0.99) is then not composited, so a finished
    # pixel can keep T as large as 0.01.  Per shard the rule restarts (the back slab begins at T = 1), so the sharded
    # picture additionally holds whatever lies behind the global stop, weighted by at most the T the single render
    # stopped with.  Everything else is fp32 rounding.
    # beyond fp32 rounding the renders may differ only where the single render took the T < 1e-4 stop (such a pixel ends
    # with 1e-4 <= T < 1e-2); elsewhere the composite must agree to rounding (no O(1) slack for unsaturated pixels)
    slack = torch.where(T_whole < 1e-2, T_whole * (float(bg.abs().max()) + float(s.features.abs().max())) * 1.001,
                        torch.zeros_like(T_whole))[None] + 1e-5 * scale
    assert bool(((comp - whole).abs() <= slack).all())
    # ... and where no pixel ever reaches the stop rule (thin scene) the composite IS the single render, to rounding
    thin = s._replace(opacities=s.opacities * 0.02)
    n2, whole2, _, _, _, img2, _ = _forward(thin, c, thin.features, bg, W, H)
    assert float(raster.image_views(img2, W, H)["final_T"].min()) > 1e-3
    parts2 = []
    for mask in (depth < cut, depth >= cut):
        A, T, _ = raster.render_partial(thin.means3D[mask], thin.features[mask], thin.opacities[mask], thin.scales[mask],
                                        thin.rotations[mask], c.world_view_transform, c.full_proj_transform,
                                        c.tanfovx, c.tanfovy, H, W, c.camera_center)
        parts2.append((A, T))
    comp2, _ = sdist.composite_over(parts2, bg)
    assert float((comp2 - whole2).abs().max()) <= 1e-4 * float(whole2.abs().max())
    # a wrong order is NOT the same picture ("over" does not commute)
    swapped, _ = sdist.composite_over(partials[::-1], bg)
    assert float((swapped - whole).abs().max()) > 1e-2 * scale


def test_cfg3_default_arithmetic_directly_against_the_oracle(orc):
    """All 512 channels on four tile rows of the headline frame, DEFAULT arithmetic (six bf16 products of exact three-term
    splits, fp32 accumulate) against the oracle AND against the exact composite (float64 sums of the same fp32 weights).

    Read element-wise -- |err| <= 1e-4 max(|x|, 1e-3 ||pixel||_inf) -- NO fp32 evaluation order but the oracle's own
    matches the oracle everywhere: elements that cancel to ~1e-3 of their pixel carry the chain's own rounding
    (K contributions x 2^-24) at the 1e-4 level.  So the test states both halves: (a) against the oracle the default is
    inside 5e-6 of the pixel's largest channel and the element-wise form fails on <= 1e-5 of the elements; (b) against
    the exact composite it is AS ACCURATE AS THE ORACLE'S fp32 chain, by the same element-wise measure."""
    from sgs_hip import raster
    from sgs_hip.synthetic import CONFIGS, make_config
    P, C, W, H, fx = CONFIGS["cfg3"]
    scene, cam = make_config("cfg3")
    pre, binn = _oracle_front(orc, scene, cam, W, H)
    s, c = scene.to(DEV), cam.to(DEV)
    out = _forward(s, c, s.features, s.bg, W, H)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    feats = scene.features.numpy()
    worst_pix, worst_elem, frac = 0.0, 0.0, 0.0
    hip_t = {"max": 0.0, "frac": 0.0, "rms": 0.0}
    orc_t = {"max": 0.0, "frac": 0.0, "rms": 0.0}
    for row in (3, 20, 41, gy - 1):
        ob = orc.blend_forward(pre, binn, feats, scene.bg.numpy(), W, H, tile_lo=row * gx, tile_hi=(row + 1) * gx)
        truth = orc.blend_forward_f64(pre, binn, feats, scene.bg.numpy(), W, H, tile_lo=row * gx, tile_hi=(row + 1) * gx)
        rows = slice(row * 16, min(H, row * 16 + 16))
        want = ob["out"][:, rows].astype(np.float64)
        got = out[1][:, rows].cpu().numpy().astype(np.float64)
        tr = truth[:, rows]
        pix_inf = np.abs(want).max(0, keepdims=True)                           # ||pixel||_inf over the channels
        err = np.abs(got - want)
        rel_elem = err / np.maximum(np.abs(want), 1e-3 * pix_inf + 1e-30)      # element-wise, floored at 1e-3 of the pixel
        worst_pix = max(worst_pix, float((err / np.maximum(pix_inf, 1e-30)).max()))
        worst_elem = max(worst_elem, float(rel_elem.max()))
        frac = max(frac, float((rel_elem > 1e-4).mean()))
        for d, x in ((hip_t, got), (orc_t, want)):
            re = np.abs(x - tr) / np.maximum(np.abs(tr), 1e-3 * pix_inf + 1e-30)
            d["max"] = max(d["max"], float(re.max()))
            d["frac"] = max(d["frac"], float((re > 1e-4).mean()))
            d["rms"] = max(d["rms"], float(np.sqrt((re ** 2).mean())))
    print(f"\ncfg3 default arithmetic, 4 tile rows x 512 channels.  vs oracle: max |err| / ||pixel||_inf = {worst_pix:.2e}, "
          f"element-wise max {worst_elem:.2e}, fraction above 1e-4: {frac:.2e}.  vs the exact composite (element-wise): "
          f"HIP max {hip_t['max']:.2e} rms {hip_t['rms']:.2e} frac>1e-4 {hip_t['frac']:.2e} | "
          f"oracle fp32 chain max {orc_t['max']:.2e} rms {orc_t['rms']:.2e} frac>1e-4 {orc_t['frac']:.2e}")
    assert worst_pix <= 5e-6          # every element within 5e-6 of its pixel's largest channel (round 2's split: 2.4e-5)
    assert worst_elem <= 1e-3 and frac <= 1e-5
    # as accurate as the reference's own arithmetic
    assert hip_t["max"] <= 1.5 * orc_t["max"] + 1e-6
    assert hip_t["rms"] <= 1.25 * orc_t["rms"] + 1e-9
    assert hip_t["frac"] <= 1.5 * orc_t["frac"] + 1e-6


def test_cfg5_full_size_on_one_gpu():
    """BASELINE config 5 AT ITS STATED SIZE on one device (288 GB HBM: the 51 GB feature table fits): 50M Gaussians x 256
    channels, 968x1296.  (a) the single render: num_rendered ~ 0.8 G instances -- 64-bit byte offsets in every binning
    array, a ~20 GB binning buffer; (b) the Gaussian-sharded data flow: eight view-space depth slabs -> eight HIP (A, T)
    partials -> ONE composite kernel (sgs_composite_over), against (a) on sampled tile rows, within the analytic slack of
    the per-shard stop rule (see the two-slab test above)."""
    from sgs_hip import raster, dist as sdist
    from sgs_hip.synthetic import CONFIGS, make_scene
    from sgs_hip.camera import pinhole
    P, C, W, H, fx = CONFIGS["cfg5"]
    free, total = torch.cuda.mem_get_info(DEV)
    if free < 150e9:
        pytest.skip(f"needs ~150 GB of free device memory (have {free / 1e9:.0f} GB)")
    scene = make_scene(P, C, W, H, fx, seed=5, features=False)     # geometry on the host (2.2 GB), features on the device
    cam = pinhole(W, H, fx)
    s, c = scene.to(DEV), cam.to(DEV)
    g = torch.Generator(device=DEV).manual_seed(55)
    feats = torch.empty(P, C, device=DEV)
    for i in range(0, P, 1 << 21):
        f = torch.randn(min(P, i + (1 << 21)) - i, C, device=DEV, generator=g)
        feats[i:i + f.shape[0]] = f / f.norm(dim=1, keepdim=True)
    del f
    bg = torch.linspace(0.0, 1.0, C, device=DEV)
    pool = raster.ScratchPool()
    n, whole, radii, _, binn, img, _ = _forward(s, c, feats, bg, W, H, pool=pool)
    assert n > 500_000_000 and int((radii > 0).sum()) > 30_000_000      # ~0.8 G instances: 4 n > 2^31 bytes per array
    iv = raster.image_views(img, W, H)
    T_whole = iv["final_T"].clone()
    r = iv["ranges"].to(torch.int64)
    assert int(r[-1, 1]) == n and bool((r[1:, 0] == r[:-1, 1]).all())    # the lists tile [0, n) exactly, in tile order
    assert float(T_whole.max()) < 1.0 and bool(torch.isfinite(whole).all())
    whole = whole.clone()
    # the same frame again: deterministic at this size too
    assert torch.equal(_forward(s, c, feats, bg, W, H, pool=pool)[1], whole)
    del binn, img, iv
    # ---- eight depth slabs (camera at the origin looking down +z: view depth = z)
    order = torch.argsort(s.means3D[:, 2])
    partials = []
    for k in range(8):
        idx = order[k * (P // 8):(k + 1) * (P // 8) if k < 7 else P]
        A, T, rad = raster.render_partial(s.means3D[idx], feats[idx], s.opacities[idx], s.scales[idx], s.rotations[idx],
                                          c.world_view_transform, c.full_proj_transform, c.tanfovx, c.tanfovy, H, W,
                                          c.camera_center, pool=pool)
        partials.append((A, T))
    comp, t_total = sdist.composite_over(partials, bg)
    scale = float(whole.abs().max())
    slack = torch.where(T_whole < 1e-2, T_whole * (float(bg.abs().max()) + 1.0) * 1.001, torch.zeros_like(T_whole))[None] + 1e-5 * scale
    for row in (2, 30, 59):
        rows = slice(row * 16, row * 16 + 16)
        assert bool(((comp[:, rows] - whole[:, rows]).abs() <= slack[:, rows]).all())
    # the composite kernel against the torch chain on one band (bit for bit: the same operations in the same order)
    band = slice(480, 496)
    ref, t_ref = partials[0][0][:, band].clone(), partials[0][1][band].clone()
    for A, T in partials[1:]:
        ref += t_ref.unsqueeze(0) * A[:, band]
        t_ref = t_ref * T[band]
    ref += bg.reshape(-1, 1, 1) * t_ref.unsqueeze(0)
    assert torch.equal(comp[:, band], ref) and torch.equal(t_total[band], t_ref)

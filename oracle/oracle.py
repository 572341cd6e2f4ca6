"""ctypes/numpy front-end of the CPU oracle (oracle/sgs_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product package never imports this module.
Parity status: "parity unpinned" (see sgs_oracle.c header and DESIGN.md).

The functions mirror the stages of the reference's Rasterizer::forward /
::backward (CR/cuda_rasterizer/rasterizer_impl.cu:198-441) and SimpleKNN::knn
(SK/simple_knn.cu:186-220) and return every intermediate so that the HIP path
can be compared stage by stage.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f = np.float32


def build(force=False):
    so = os.path.join(_HERE, "libsgs_oracle.so")
    src = os.path.join(_HERE, "sgs_oracle.c")
    srcs = (src, os.path.join(_HERE, "sh_poly_table_c.h"))
    if force or not os.path.exists(so) or any(os.path.getmtime(so) < os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libsgs_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_expf.restype = C.c_float
        _LIB.orc_expf.argtypes = [C.c_float]
        _LIB.orc_higher_msb.restype = C.c_uint32
        _LIB.orc_higher_msb.argtypes = [C.c_uint32]
        _LIB.orc_prep_morton.restype = C.c_uint32
        _LIB.orc_prep_morton.argtypes = [C.c_uint32]
        _LIB.orc_inclusive_scan.restype = C.c_uint32
        _LIB.orc_preprocess.restype = C.c_int
    return _LIB


def _p(a):
    """numpy array (or None) -> void* ; None -> NULL ("not provided")."""
    if a is None:
        return C.c_void_p(0)
    assert a.flags["C_CONTIGUOUS"]
    return C.c_void_p(a.ctypes.data)


def _c32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def expf(x):
    x = np.asarray(x, dtype=np.float32)
    out = np.empty_like(x)
    L = lib()
    flat_in, flat_out = x.ravel(), out.ravel()
    for i in range(flat_in.size):
        flat_out[i] = L.orc_expf(float(flat_in[i]))
    return out


def higher_msb(n):
    return int(lib().orc_higher_msb(int(n)))


def prep_morton(x):
    return int(lib().orc_prep_morton(int(x)))


def tile_grid(W, H):
    return (W + 15) // 16, (H + 15) // 16


def preprocess(means3D, opacities, view, proj, campos, W, H, tanfovx, tanfovy, scales=None,
               rotations=None, scale_modifier=1.0, cov3D_precomp=None, colors_precomp=None,
               shs=None, sh_degree=0, prefiltered=False, num_channels=3):
    means3D = _c32(means3D)
    P = means3D.shape[0]
    opacities = _c32(opacities).reshape(-1)
    scales, rotations = _c32(scales), _c32(rotations)
    cov3D_precomp, colors_precomp, shs = _c32(cov3D_precomp), _c32(colors_precomp), _c32(shs)
    view, proj, campos = _c32(view).reshape(16), _c32(proj).reshape(16), _c32(campos).reshape(3)
    M = 0 if shs is None else shs.shape[1]
    o = dict(
        radii=np.zeros(P, np.int32), means2D=np.zeros((P, 2), _f), depths=np.zeros(P, _f),
        cov3D=np.zeros((P, 6), _f), rgb=np.zeros((P, 3), _f), clamped=np.zeros((P, 3), np.uint8),
        conic_opacity=np.zeros((P, 4), _f), tiles_touched=np.zeros(P, np.uint32))
    rc = lib().orc_preprocess(
        C.c_int(P), C.c_int(sh_degree), C.c_int(M), _p(means3D), _p(scales),
        C.c_float(scale_modifier), _p(rotations), _p(opacities), _p(shs), _p(cov3D_precomp),
        _p(colors_precomp), _p(view), _p(proj), _p(campos), C.c_int(W), C.c_int(H),
        C.c_float(tanfovx), C.c_float(tanfovy), C.c_int(int(prefiltered)), C.c_int(num_channels),
        _p(o["radii"]), _p(o["means2D"]), _p(o["depths"]), _p(o["cov3D"]), _p(o["rgb"]),
        _p(o["clamped"]), _p(o["conic_opacity"]), _p(o["tiles_touched"]))
    if rc != 0:
        raise RuntimeError("Point is filtered although prefiltered is set.")
    return o


def mark_visible(means3D, view, proj):
    means3D = _c32(means3D)
    P = means3D.shape[0]
    out = np.zeros(P, np.uint8)
    lib().orc_mark_visible(C.c_int(P), _p(means3D), _p(_c32(view).reshape(16)),
                           _p(_c32(proj).reshape(16)), _p(out))
    return out.astype(bool)


def binning(pre, W, H):
    """scan -> duplicateWithKeys -> stable radix sort -> tile ranges."""
    P = pre["radii"].shape[0]
    gx, gy = tile_grid(W, H)
    offsets = np.zeros(P, np.uint32)
    L = int(lib().orc_inclusive_scan(C.c_int(P), _p(pre["tiles_touched"]), _p(offsets))) if P else 0
    keys_u = np.zeros(L, np.uint64)
    vals_u = np.zeros(L, np.uint32)
    lib().orc_duplicate_with_keys(C.c_int(P), _p(pre["means2D"]), _p(pre["depths"]), _p(offsets),
                                  _p(pre["radii"]), C.c_int(W), C.c_int(H), _p(keys_u), _p(vals_u))
    bit = higher_msb(gx * gy)
    keys_s = np.zeros(L, np.uint64)
    vals_s = np.zeros(L, np.uint32)
    lib().orc_sort_pairs(C.c_size_t(L), _p(keys_u), _p(vals_u), _p(keys_s), _p(vals_s),
                         C.c_int(32 + bit))
    ranges = np.zeros((gx * gy, 2), np.uint32)
    lib().orc_tile_ranges(C.c_size_t(L), _p(keys_s), C.c_int(gx * gy), _p(ranges))
    return dict(point_offsets=offsets, num_rendered=L, keys_unsorted=keys_u, vals_unsorted=vals_u,
                keys_sorted=keys_s, point_list=vals_s, ranges=ranges, sort_bits=32 + bit)


def blend_forward(pre, binn, features, bg, W, H, want_depth=False, tile_lo=None, tile_hi=None):
    features = _c32(features)
    Cn = features.shape[1] if features.ndim == 2 else 0
    bg = _c32(bg).reshape(-1)
    assert bg.shape[0] >= Cn, "bg shorter than num_channels"
    out = np.zeros((Cn, H, W), _f)
    final_T = np.zeros((H, W), _f)
    n_contrib = np.zeros((H, W), np.uint32)
    depth = np.zeros((1, H, W), _f) if want_depth else None
    gx, gy = tile_grid(W, H)
    lo = 0 if tile_lo is None else tile_lo
    hi = gx * gy if tile_hi is None else tile_hi
    lib().orc_blend_forward_tiles(
        C.c_int(W), C.c_int(H), C.c_int(Cn), _p(binn["ranges"]), _p(binn["point_list"]),
        _p(pre["means2D"]), _p(features), _p(pre["conic_opacity"]), _p(pre["depths"]), _p(bg),
        _p(out), _p(final_T), _p(n_contrib), _p(depth), C.c_int(lo), C.c_int(hi))
    return dict(out=out, final_T=final_T, n_contrib=n_contrib, depth=depth)


def blend_forward_f64(pre, binn, features, bg, W, H, tile_lo=None, tile_hi=None):
    """The composite with the channel sums in float64 (weights and every decision stay the contract's fp32 values):
    the exact value the fp32 multiply-add chain approximates.  (C,H,W) float64; tiles outside [lo, hi) stay 0."""
    features = _c32(features)
    Cn = features.shape[1]
    bg = _c32(bg).reshape(-1)
    assert bg.shape[0] >= Cn, "bg shorter than num_channels"
    out = np.zeros((Cn, H, W), np.float64)
    gx, gy = tile_grid(W, H)
    lo = 0 if tile_lo is None else tile_lo
    hi = gx * gy if tile_hi is None else tile_hi
    lib().orc_blend_forward_tiles_f64(
        C.c_int(W), C.c_int(H), C.c_int(Cn), _p(binn["ranges"]), _p(binn["point_list"]),
        _p(pre["means2D"]), _p(features), _p(pre["conic_opacity"]), _p(bg), _p(out), C.c_int(lo), C.c_int(hi))
    return out


def forward(means3D, opacities, view, proj, campos, W, H, tanfovx, tanfovy, bg, num_channels,
            scales=None, rotations=None, scale_modifier=1.0, cov3D_precomp=None,
            colors_precomp=None, shs=None, sh_degree=0, prefiltered=False, want_depth=False):
    """Whole Rasterizer::forward (CR/rasterizer_impl.cu:198-341; RR adds depth)."""
    if num_channels != 3 and colors_precomp is None:
        raise RuntimeError("For non-RGB, provide precomputed Gaussian colors!")
    P = np.asarray(means3D).shape[0]
    if P == 0:  # CR/rasterize_points.cu:85-120: zeros, not bg
        return dict(out=np.zeros((num_channels, H, W), _f), radii=np.zeros(0, np.int32),
                    num_rendered=0, depth=np.zeros((1, H, W), _f) if want_depth else None)
    pre = preprocess(means3D, opacities, view, proj, campos, W, H, tanfovx, tanfovy, scales,
                     rotations, scale_modifier, cov3D_precomp, colors_precomp, shs, sh_degree,
                     prefiltered, num_channels)
    binn = binning(pre, W, H)
    feats = _c32(colors_precomp) if colors_precomp is not None else pre["rgb"]
    bl = blend_forward(pre, binn, feats, bg, W, H, want_depth)
    r = dict(pre)
    r.update(binn)
    r.update(bl)
    r["features"] = feats
    return r


def blend_backward(fwd, bg, dL_dout, W, H, tile_lo=None, tile_hi=None):
    """tile_lo / tile_hi: only those tiles' pixels are walked (= the backward for a dL_dout that is zero elsewhere; final_T and
    n_contrib need only be valid there)."""
    feats = fwd["features"]
    P, Cn = feats.shape
    dL_dout = _c32(dL_dout)
    bg = _c32(bg).reshape(-1)
    g = dict(dL_dmean2D=np.zeros((P, 3), _f), dL_dconic=np.zeros((P, 4), _f),
             dL_dopacity=np.zeros((P, 1), _f), dL_dcolors=np.zeros((P, Cn), _f))
    lib().orc_blend_backward(
        C.c_int(P), C.c_int(W), C.c_int(H), C.c_int(Cn), _p(fwd["ranges"]), _p(fwd["point_list"]),
        _p(bg), _p(fwd["means2D"]), _p(fwd["conic_opacity"]), _p(feats), _p(fwd["final_T"]),
        _p(fwd["n_contrib"]), _p(dL_dout), _p(g["dL_dmean2D"]), _p(g["dL_dconic"]),
        _p(g["dL_dopacity"]), _p(g["dL_dcolors"]), C.c_int(0 if tile_lo is None else tile_lo),
        C.c_int(((W + 15) // 16) * ((H + 15) // 16) if tile_hi is None else tile_hi))
    return g


def backward(fwd, dL_dout, means3D, view, proj, campos, W, H, tanfovx, tanfovy, bg, scales=None,
             rotations=None, scale_modifier=1.0, cov3D_precomp=None, shs=None, sh_degree=0, tile_lo=None, tile_hi=None):
    """Whole Rasterizer::backward (CR/rasterizer_impl.cu:345-441), runtime C."""
    means3D = _c32(means3D)
    P = means3D.shape[0]
    g = blend_backward(fwd, bg, dL_dout, W, H, tile_lo, tile_hi)
    scales, rotations, shs = _c32(scales), _c32(rotations), _c32(shs)
    M = 0 if shs is None else shs.shape[1]
    cov3Ds = _c32(cov3D_precomp) if cov3D_precomp is not None else fwd["cov3D"]
    g.update(dL_dmeans3D=np.zeros((P, 3), _f), dL_dcov3D=np.zeros((P, 6), _f),
             dL_dsh=np.zeros((P, M, 3), _f), dL_dscales=np.zeros((P, 3), _f),
             dL_drotations=np.zeros((P, 4), _f))
    fy = np.float32(H) / (np.float32(2.0) * np.float32(tanfovy))
    fx = np.float32(W) / (np.float32(2.0) * np.float32(tanfovx))
    clamped = np.ascontiguousarray(fwd["clamped"])
    lib().orc_preprocess_backward(
        C.c_int(P), C.c_int(sh_degree), C.c_int(M), _p(means3D), _p(fwd["radii"]), _p(shs),
        _p(clamped), _p(scales), _p(rotations), C.c_float(scale_modifier), _p(cov3Ds),
        _p(_c32(view).reshape(16)), _p(_c32(proj).reshape(16)), C.c_float(fx), C.c_float(fy),
        C.c_float(tanfovx), C.c_float(tanfovy), _p(_c32(campos).reshape(3)), _p(g["dL_dmean2D"]),
        _p(g["dL_dconic"]), _p(g["dL_dmeans3D"]), _p(g["dL_dcolors"]), _p(g["dL_dcov3D"]),
        _p(g["dL_dsh"]), _p(g["dL_dscales"]), _p(g["dL_drotations"]))
    return g


def dist2(points):
    """distCUDA2 (SK/spatial.cu:15-26): mean squared distance to the 3 nearest others."""
    pts = _c32(points)
    P = pts.shape[0]
    out = np.zeros(P, _f)
    lib().orc_dist2_bruteforce(C.c_int(P), _p(pts), _p(out))
    return out

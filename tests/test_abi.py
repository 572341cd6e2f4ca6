"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/sgs_raster.h declares, host-only entry points behave, and the Python mirror enforces
the reference's argument contract without a GPU (no compute calls here)."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "sgs_raster.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(sgs_[a-z0-9_]+)\s*\(", hdr)) - {"sgs_alloc_fn"})


def test_library_exports_every_declared_symbol():
    from sgs_hip import _lib
    lib = _lib.load()          # raises ImportError if the HIP extension is missing
    syms = _declared_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(lib, s), s
    assert set(syms) == set(_lib.EXPORTS)
    assert lib.sgs_abi_version() == 1


def test_host_only_entry_points():
    from sgs_hip import _lib, raster
    lib = _lib.load()
    assert raster.sort_bits(1296, 968) == 45      # 81*61 = 4941 tiles -> 13 bits
    assert raster.sort_bits(256, 256) == 41
    assert raster.sort_bits(16, 16) == 33
    lay = _lib.ImageLayout()
    assert lib.sgs_image_layout_of(1296, 968, C.byref(lay)) == 0
    assert lay.accum_alpha % 128 == 0 and lay.n_contrib % 128 == 0 and lay.ranges % 128 == 0
    assert lay.n_contrib >= 1296 * 968 * 4 and lay.total >= lay.ranges + 4941 * 8
    assert lib.sgs_image_layout_of(-1, 5, C.byref(lay)) < 0
    assert "bad" in _lib.last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from sgs_hip import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()


def test_argument_contract_errors_match_reference():
    import channel_rasterization as chn
    import rgbd_rasterization as rgbd
    z = torch.zeros(4, 3)
    s = chn.GaussianRasterizationSettings(
        image_height=16, image_width=16, tanfovx=1.0, tanfovy=1.0, bg=torch.zeros(3),
        scale_modifier=1.0, viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=0,
        campos=torch.zeros(3), prefiltered=False, debug=False, num_channels=3)
    r = chn.GaussianRasterizer(raster_settings=s)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(means3D=z, means2D=z, opacities=z[:, :1], scales=z, rotations=torch.zeros(4, 4))
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(means3D=z, means2D=z, opacities=z[:, :1], shs=torch.zeros(4, 16, 3), colors_precomp=z,
          scales=z, rotations=torch.zeros(4, 4))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=z, means2D=z, opacities=z[:, :1], colors_precomp=z, scales=z)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=z, means2D=z, opacities=z[:, :1], colors_precomp=z, scales=z,
          rotations=torch.zeros(4, 4), cov3D_precomp=torch.zeros(4, 6))
    # CPU tensors are rejected: the product has no CPU path
    with pytest.raises(RuntimeError, match="no CPU path"):
        r(means3D=z, means2D=z, opacities=z[:, :1], colors_precomp=z, scales=z,
          rotations=torch.zeros(4, 4))
    with pytest.raises(RuntimeError, match=r"\(num_points, 3\)"):
        r(means3D=torch.zeros(4, 2), means2D=z, opacities=z[:, :1], colors_precomp=z, scales=z,
          rotations=torch.zeros(4, 4))
    # settings tuples have the reference's fields, in order
    assert chn.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix",
        "projmatrix", "sh_degree", "campos", "prefiltered", "debug", "num_channels")
    assert rgbd.GaussianRasterizationSettings._fields == chn.GaussianRasterizationSettings._fields[:-1]
    from simple_knn._C import distCUDA2
    with pytest.raises(RuntimeError, match="no CPU path"):
        distCUDA2(torch.zeros(5, 3))


def test_one_shot_stream_options_do_not_survive_a_call_that_returns_early():
    """SGS_OPT_OUT_PITCH / SGS_OPT_NORM_PLANE belong to the NEXT forward on the stream whatever becomes of it.
    A forward that returns early -- P == 0 (rasterize_points.cu:85-120: the caller returns zeros) or an argument error --
    must consume them too, or the following forward would write a contiguous buffer with a stale pitch.  Host-only: the
    early returns happen before any device work."""
    from sgs_hip import _lib
    lib = _lib.load()
    NONE = 0x7fffffff   # sgs_stream_set_option's "there was no override"

    @_lib.ALLOC_FN
    def alloc(user, n):
        return None
    out = (C.c_float * 4)()

    def forward(P, with_callbacks):
        cb = alloc if with_callbacks else _lib.ALLOC_FN()
        return lib.sgs_rasterize_forward(cb, None, cb, None, cb, None, P, 0, 0, None, 16, 16, None, None, C.addressof(out), None, None,
                                         1.0, None, None, None, None, None, 1.0, 1.0, 0, 128, C.addressof(out), None, None,
                                         0, None)

    for P, with_callbacks, want_rc in ((0, True, 0), (5, False, "error"), (-1, True, "error")):
        assert lib.sgs_stream_set_option(None, _lib.OPT_OUT_PITCH, 1312) == NONE
        assert lib.sgs_stream_set_option(None, _lib.OPT_NORM_PLANE, 1) == NONE
        rc = forward(P, with_callbacks)
        assert (rc == 0) if want_rc == 0 else (rc < 0), (P, with_callbacks, rc)
        # clearing returns the previous override: there must be none left
        assert lib.sgs_stream_set_option(None, _lib.OPT_OUT_PITCH, -1) == NONE, "stale output pitch survived an early return"
        assert lib.sgs_stream_set_option(None, _lib.OPT_NORM_PLANE, -1) == NONE, "stale norm-plane flag survived an early return"

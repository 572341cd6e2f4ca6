"""Gaussian PLY and fusion .pt I/O (model/gaussian_model.py:250-344, fusion.py:234-257).
Parity unpinned against a reference-WRITTEN file (the reference's reader / writer needs `plyfile`, absent here); the layout its
code defines is pinned by a hand-derived byte fixture (tests/golden/gaussians_deg1.ply): header text against the PLY 1.0 layout the
reference's attribute list implies, round trips, and reading by NAME like load_ply."""
import os
import struct

import numpy as np
import pytest
import torch

from sgs_hip import io as sio


def _model(P=37, deg=3, seed=0):
    g = torch.Generator().manual_seed(seed)
    K = (deg + 1) ** 2 - 1
    return dict(xyz=torch.randn(P, 3, generator=g), features_dc=torch.randn(P, 1, 3, generator=g),
                features_rest=torch.randn(P, K, 3, generator=g), opacity=torch.randn(P, 1, generator=g),
                scaling=torch.randn(P, 3, generator=g), rotation=torch.randn(P, 4, generator=g))


@pytest.mark.parametrize("deg", [0, 1, 3])
def test_gaussian_ply_round_trip_and_header(tmp_path, deg):
    m = _model(deg=deg)
    path = str(tmp_path / "point_cloud.ply")
    sio.write_gaussian_ply(path, **m)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode("ascii").split("\n")[:-1]
    n_rest = 3 * ((deg + 1) ** 2 - 1)
    want = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(n_rest)] + \
           ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 37"]
    assert lines[3:] == [f"property float {n}" for n in want]
    assert len(body) == 37 * 4 * len(want)
    # first record, field by field: channel-major SH (the reference's transpose(1, 2).flatten)
    rec = struct.unpack("<" + "f" * len(want), body[:4 * len(want)])
    assert rec[0:3] == tuple(m["xyz"][0].tolist()) and rec[3:6] == (0.0, 0.0, 0.0)
    assert rec[6:9] == tuple(m["features_dc"][0, 0].tolist())
    if n_rest:
        assert rec[9:9 + n_rest] == tuple(m["features_rest"][0].t().reshape(-1).tolist())
    back = sio.read_gaussian_ply(path, max_sh_degree=deg)
    for k, v in m.items():
        assert back[k].dtype == torch.float32 and torch.equal(back[k], v), k


def test_reader_picks_properties_by_name(tmp_path):
    """A file with shuffled property order, an extra double property and a second element, as another tool
    might write it."""
    m = _model(P=5, deg=0)
    names = ["rot_3", "rot_2", "rot_1", "rot_0", "opacity", "z", "y", "x", "nx", "ny", "nz", "f_dc_2", "f_dc_1",
             "f_dc_0", "scale_2", "scale_1", "scale_0"]
    cols = {"x": m["xyz"][:, 0], "y": m["xyz"][:, 1], "z": m["xyz"][:, 2], "nx": torch.zeros(5), "ny": torch.zeros(5),
            "nz": torch.zeros(5), "opacity": m["opacity"][:, 0]}
    for i in range(3):
        cols[f"f_dc_{i}"] = m["features_dc"][:, 0, i]
        cols[f"scale_{i}"] = m["scaling"][:, i]
    for i in range(4):
        cols[f"rot_{i}"] = m["rotation"][:, i]
    header = ["ply", "format binary_big_endian 1.0", "comment made by hand", "element vertex 5", "property double confidence"]
    header += [f"property float {n}" for n in names] + ["element face 0", "property list uchar int vertex_indices", "end_header"]
    path = str(tmp_path / "other.ply")
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode())
        for r in range(5):
            f.write(struct.pack(">d", 0.5))
            f.write(struct.pack(">" + "f" * len(names), *[float(cols[n][r]) for n in names]))
    back = sio.read_gaussian_ply(path, max_sh_degree=0)
    for k in ("xyz", "features_dc", "opacity", "scaling", "rotation"):
        assert torch.equal(back[k], m[k]), k
    assert back["features_rest"].shape == (5, 0, 3)
    with pytest.raises(ValueError):
        sio.read_gaussian_ply(path, max_sh_degree=3)     # the reference asserts the f_rest count too


def test_fusion_pt_round_trip(tmp_path):
    g = torch.Generator().manual_seed(1)
    feat = torch.randn(50, 16, generator=g)
    p1, p2 = str(tmp_path / "0.pt"), str(tmp_path / "1.pt")
    sio.save_fusion_features(p1, feat)
    d = torch.load(p1)
    assert d["feat"].dtype == torch.float16 and d["feat"].shape == (50, 16)
    assert d["mask_full"].dtype == torch.bool and bool(d["mask_full"].all())
    mask = torch.rand(50, generator=g) > 0.5
    sio.save_fusion_features(p2, feat, mask)
    d = torch.load(p2)
    assert d["feat"].shape == (int(mask.sum()), 16) and torch.equal(d["mask_full"], mask)
    back, m2 = sio.load_fusion_features(p2)
    assert torch.equal(m2, mask) and torch.equal(back[mask], feat[mask].half().float()) and float(back[~mask].abs().sum()) == 0


def test_fusion_pt_fixture_in_the_reference_format(tmp_path):
    """The fusion `.pt` format pinned the way the PLY is: tests/golden/gen_fusion_pt_fixture.py writes what fusion.py:234-257 writes
    (torch.save of {"feat": fp16 rows, "mask_full": bool}; masked files hold only the masked rows) with torch alone; the reader
    returns the closed-form values with zeros for unmasked points, and the writer's files load into dicts equal to the fixtures'
    (keys in the same order, dtypes, shapes, values).  Not a file written BY the reference (it needs omegaconf / a scene) -- its
    two torch.save statements are what the generator executes."""
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    P, C = 12, 8
    want = (torch.arange(P, dtype=torch.float32)[:, None] - 3.0) / 4.0 + torch.arange(C, dtype=torch.float32)[None, :] / 64.0
    sel = torch.zeros(P, dtype=torch.bool)
    sel[[1, 4, 5, 9, 11]] = True
    for name, mask in (("fusion_full.pt", None), ("fusion_masked.pt", sel)):
        feat, m = sio.load_fusion_features(os.path.join(gold, name))
        ref_mask = torch.ones(P, dtype=torch.bool) if mask is None else mask
        assert m.dtype == torch.bool and torch.equal(m, ref_mask)
        assert feat.dtype == torch.float32 and torch.equal(feat[ref_mask], want[ref_mask]) and float(feat[~ref_mask].abs().sum()) == 0
        out = str(tmp_path / name)
        sio.save_fusion_features(out, want, mask)
        a, b = torch.load(out), torch.load(os.path.join(gold, name))
        assert list(a.keys()) == list(b.keys()) == ["feat", "mask_full"]
        for k in a:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and torch.equal(a[k], b[k]), (name, k)


def test_ascii_ply_is_read_too(tmp_path):
    m = _model(P=3, deg=0)
    names = sio.gaussian_attribute_names(3, 0)
    rows = torch.cat([m["xyz"], torch.zeros(3, 3), m["features_dc"][:, 0, :], m["opacity"], m["scaling"], m["rotation"]], dim=1)
    path = str(tmp_path / "ascii.ply")
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex 3\n" + "".join(f"property float {n}\n" for n in names) + "end_header\n")
        for r in rows.tolist():
            f.write(" ".join(repr(float(v)) for v in r) + "\n")
    back = sio.read_gaussian_ply(path, max_sh_degree=0)
    for k in ("xyz", "features_dc", "opacity", "scaling", "rotation"):
        assert torch.equal(back[k], m[k]), k


def test_hand_derived_ply_fixture_reads_and_rewrites_byte_for_byte(tmp_path):
    """tests/golden/gaussians_deg1.ply was written with `struct` alone by tests/golden/gen_ply_fixture.py from the reference's
    save_ply / construct_list_of_attributes (model/gaussian_model.py:250-281: attribute order, f4 everywhere, zero normals,
    transpose(1, 2).flatten of the SH blocks) -- it pins the LAYOUT that code defines; it is NOT a file the reference wrote
    (plyfile is not in this image).  The reader must return the values the generator put in (load_ply's reshape,
    :288-344), and the writer must reproduce the file byte for byte."""
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    want = np.load(os.path.join(gold, "gaussians_deg1_values.npz"))
    path = os.path.join(gold, "gaussians_deg1.ply")
    back = sio.read_gaussian_ply(path, max_sh_degree=1)
    for k in ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation"):
        assert back[k].dtype == torch.float32 and np.array_equal(back[k].numpy(), want[k]), k
    out = str(tmp_path / "again.ply")
    sio.write_gaussian_ply(out, **{k: torch.from_numpy(want[k]) for k in want.files})
    assert open(out, "rb").read() == open(path, "rb").read()

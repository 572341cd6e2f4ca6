"""Hand-derived Gaussian PLY fixture.  `plyfile` -- which the reference's save_ply / load_ply
(/root/reference/model/gaussian_model.py:250-281, 288-344) go through -- is not in this image, so NO reference-written file
can be produced; this script writes, with `struct` only and WITHOUT touching sgs_hip.io, the bytes that code path defines:

  * construct_list_of_attributes (:250-263): x y z nx ny nz, f_dc_0..2, f_rest_0..(3 K - 1), opacity, scale_0..2, rot_0..3;
  * save_ply (:265-281): every attribute an "f4", normals zero, f_dc / f_rest = `_features_*.transpose(1, 2).flatten(1)`
    (CHANNEL-major: f_rest_{c K + k} = features_rest[p, k, c]), one numpy record per Gaussian, written by
    PlyElement.describe(elements, "vertex") + PlyData([el]).write(path): plyfile's binary default on a little-endian host
    = "format binary_little_endian 1.0", one "property float <name>" line per f4 field, no comment lines.

The fixture pins the LAYOUT the reference's code defines, not a file the reference wrote (tests/test_io.py says so).
    python tests/golden/gen_ply_fixture.py      -> tests/golden/gaussians_deg1.ply (+ the values as .npz)"""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
P, DEG = 5, 1
K = (DEG + 1) ** 2 - 1


def values():
    """Deterministic, exactly representable fp32 values, every field of every Gaussian distinct."""
    i = np.arange(P, dtype=np.float32)[:, None]
    xyz = i * 8 + np.array([[0.5, 1.5, 2.5]], np.float32)
    dc = (i * 8 + np.array([[10.25, 11.25, 12.25]], np.float32))[:, None, :]                       # (P, 1, 3)
    rest = i[:, :, None] * 64 + np.arange(K, dtype=np.float32)[None, :, None] * 4 + np.arange(3, dtype=np.float32)[None, None, :] + 100.0   # (P, K, 3)
    opacity = i * 0.5 - 1.0
    scaling = i * 4 + np.array([[-3.0, -2.0, -1.0]], np.float32)
    rotation = i * 2 + np.array([[1.0, 0.25, 0.5, 0.75]], np.float32)
    return dict(xyz=xyz.astype(np.float32), features_dc=dc.astype(np.float32), features_rest=rest.astype(np.float32),
                opacity=opacity.astype(np.float32), scaling=scaling.astype(np.float32), rotation=rotation.astype(np.float32))


def main():
    v = values()
    names = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(3 * K)] + \
            ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]
    head = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % P + "".join(f"property float {n}\n" for n in names) + "end_header\n"
    body = b""
    for p in range(P):
        rec = list(v["xyz"][p]) + [0.0, 0.0, 0.0]
        rec += [v["features_dc"][p, 0, c] for c in range(3)]                     # transpose(1, 2).flatten: channel-major
        rec += [v["features_rest"][p, k, c] for c in range(3) for k in range(K)]
        rec += list(v["opacity"][p]) + list(v["scaling"][p]) + list(v["rotation"][p])
        assert len(rec) == len(names)
        body += struct.pack("<%df" % len(rec), *[float(x) for x in rec])
    open(os.path.join(HERE, "gaussians_deg1.ply"), "wb").write(head.encode("ascii") + body)
    np.savez(os.path.join(HERE, "gaussians_deg1_values.npz"), **v)


if __name__ == "__main__":
    main()

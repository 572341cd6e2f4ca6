"""the on-disk formats either side of the hot path, without `plyfile`.

  * Gaussian PLY -- what GaussianModel.save_ply / load_ply write and read (model/gaussian_model.py:250-281,
    288-344): one `vertex` element, binary little endian, every property `float`:
    x y z nx ny nz f_dc_0..2 f_rest_0..(3*((deg+1)^2-1)-1) opacity scale_0..2 rot_0..3, the SH coefficients
    stored channel-major (the reference's transpose(1, 2).flatten(start_dim=1)).
  * fusion `.pt` -- {"feat": fp16 (N,C), "mask_full": bool (P)} (fusion.py:234-257).

Parity UNPINNED: the reference's own reader / writer needs `plyfile`, which this image does not have, so no
file produced by it could be generated here.  The writer emits the header `plyfile` emits for such an element
(PLY 1.0, `property float <name>` lines in the reference's attribute order); the reader parses any PLY header
(ascii or binary little / big endian scalar properties) and picks properties BY NAME the way load_ply does.
Host-side I/O only (NumPy / torch); nothing here touches the GPU."""
import numpy as np
import torch

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2",
              "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
              "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def gaussian_attribute_names(n_dc, n_rest, n_scale=3, n_rot=4):
    """construct_list_of_attributes (model/gaussian_model.py:250-262)."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(n_dc)]
    names += [f"f_rest_{i}" for i in range(n_rest)]
    names.append("opacity")
    names += [f"scale_{i}" for i in range(n_scale)]
    names += [f"rot_{i}" for i in range(n_rot)]
    return names


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def write_gaussian_ply(path, xyz, features_dc, features_rest, opacity, scaling, rotation):
    """save_ply (model/gaussian_model.py:264-281).  features_dc (P,1,3), features_rest (P,K,3) in the model's
    layout (coefficient-major); opacity (P,1); scaling (P,3); rotation (P,4); the raw (pre-activation)
    parameters, as the reference stores them."""
    xyz = _np(xyz).astype(np.float32)
    P = xyz.shape[0]
    f_dc = np.ascontiguousarray(np.transpose(_np(features_dc), (0, 2, 1))).reshape(P, -1).astype(np.float32)
    f_rest = np.ascontiguousarray(np.transpose(_np(features_rest), (0, 2, 1))).reshape(P, -1).astype(np.float32)
    cols = [xyz, np.zeros_like(xyz), f_dc, f_rest, _np(opacity).reshape(P, -1).astype(np.float32),
            _np(scaling).reshape(P, -1).astype(np.float32), _np(rotation).reshape(P, -1).astype(np.float32)]
    table = np.ascontiguousarray(np.concatenate(cols, axis=1), dtype="<f4")
    names = gaussian_attribute_names(f_dc.shape[1], f_rest.shape[1], cols[5].shape[1], cols[6].shape[1])
    assert table.shape[1] == len(names)
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {P}"]
    header += [f"property float {n}" for n in names]
    header.append("end_header")
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(table.tobytes())


def read_ply_vertices(path):
    """Header + the first element of a PLY file as a structured NumPy array (scalar properties only)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_first, seen_element = None, None, [], False, 0
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: unterminated PLY header")
            tok = line.decode("ascii").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                seen_element += 1
                in_first = seen_element == 1
                if in_first:
                    count = int(tok[2])
            elif tok[0] == "property" and in_first:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties in the first element are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt is None or count is None:
            raise ValueError(f"{path}: incomplete PLY header")
        if fmt == "ascii":
            rows = [f.readline().split() for _ in range(count)]
            out = np.empty(count, dtype=[(n, t) for n, t in props])
            for j, (n, t) in enumerate(props):
                out[n] = np.array([r[j] for r in rows], dtype=np.float64).astype(t)
            return out
        order = {"binary_little_endian": "<", "binary_big_endian": ">"}[fmt]
        dt = np.dtype([(n, order + t) for n, t in props])
        data = np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)
        return data


def read_gaussian_ply(path, max_sh_degree=3):
    """load_ply (model/gaussian_model.py:288-344): tensors in the model's layouts (float32, CPU):
    xyz (P,3), features_dc (P,1,3), features_rest (P,K,3), opacity (P,1), scaling (P,3), rotation (P,4)."""
    v = read_ply_vertices(path)
    names = v.dtype.names
    P = v.shape[0]

    def numbered(prefix):
        sel = sorted((n for n in names if n.startswith(prefix)), key=lambda n: int(n.split("_")[-1]))
        return np.stack([v[n].astype(np.float32) for n in sel], axis=1) if sel else np.zeros((P, 0), np.float32)

    xyz = np.stack([v["x"], v["y"], v["z"]], axis=1).astype(np.float32)
    dc = np.stack([v["f_dc_0"], v["f_dc_1"], v["f_dc_2"]], axis=1).astype(np.float32).reshape(P, 3, 1)
    rest = numbered("f_rest_")
    if rest.shape[1] != 3 * (max_sh_degree + 1) ** 2 - 3:
        raise ValueError(f"{path}: {rest.shape[1]} f_rest_* properties, expected {3 * (max_sh_degree + 1) ** 2 - 3} "
                         f"for max_sh_degree={max_sh_degree}")
    rest = rest.reshape(P, 3, (max_sh_degree + 1) ** 2 - 1)
    t = torch.from_numpy
    return {"xyz": t(xyz), "features_dc": t(np.ascontiguousarray(np.transpose(dc, (0, 2, 1)))),
            "features_rest": t(np.ascontiguousarray(np.transpose(rest, (0, 2, 1)))),
            "opacity": t(v["opacity"].astype(np.float32)[:, None].copy()), "scaling": t(numbered("scale_")),
            "rotation": t(numbered("rot"))}


def save_fusion_features(path, features_semantic, mask_full=None):
    """fusion.py:234-257: {"feat": fp16 (N,C), "mask_full": bool (P)}; with a mask only its rows are stored."""
    f = features_semantic.detach().cpu()
    if mask_full is None:
        mask_full = torch.ones(f.shape[0], dtype=torch.bool)
    else:
        mask_full = mask_full.detach().cpu().to(torch.bool)
        f = f[mask_full]
    torch.save({"feat": f.half(), "mask_full": mask_full}, path)


def load_fusion_features(path, device="cpu"):
    """-> (features (P,C) float32 with zeros for unmasked points, mask_full (P) bool)."""
    d = torch.load(path, map_location="cpu")
    mask = d["mask_full"].to(torch.bool)
    feat = torch.zeros(mask.shape[0], d["feat"].shape[1], dtype=torch.float32)
    feat[mask] = d["feat"].float()
    return feat.to(device), mask.to(device)

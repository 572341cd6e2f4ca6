"""the 2-D -> 3-D fusion step either side of the depth render, on the device.

`PointCloudToImageMapper` mirrors the reference class (dataset/fusion_utils.py:16-78): same constructor,
same `compute_mapping(world_to_camera, coords, depth=None, intrinsic=None)` contract (NumPy in, NumPy out:
`mapping` (N,3) int rows (y, x, 1) / (0,0,0) and `weight` (N,)), so fusion.py:127-133 runs unchanged --
but the work is one HIP kernel, and `compute_mapping_device` keeps everything in torch tensors on the GPU
(no copy of the centres / the rendered depth to the host per view).  `accumulate_features` is the
per-view scatter of fusion.py:139-147.  No CPU fallback: the HIP library is required.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class PointCloudToImageMapper(object):
    def __init__(self, image_dim, visibility_threshold=0.25, cut_bound=0, intrinsics=None, device="cuda:0"):
        self.image_dim = image_dim
        self.vis_thres = visibility_threshold
        self.cut_bound = cut_bound
        self.device = torch.device(device)
        # the constructor's rescaling of the intrinsics (fusion_utils.py:22-28), float64 like the reference
        self.intrinsics = np.array(intrinsics, dtype=np.float64).copy()
        scale_x = self.image_dim[0] / (self.intrinsics[0, 2] * 2)
        scale_y = self.image_dim[1] / (self.intrinsics[1, 2] * 2)
        self.intrinsics[0, 0] *= scale_x
        self.intrinsics[1, 1] *= scale_y
        self.intrinsics[0, 2] = self.image_dim[0] / 2
        self.intrinsics[1, 2] = self.image_dim[1] / 2

    def compute_mapping_device(self, world_to_camera, coords, depth=None, intrinsic=None):
        """Device tensors in and out.  world_to_camera: (4,4) float32 (view.world_view_transform, i.e. the
        transposed matrix, as the reference passes it); coords (N,3) float32; depth: None, an (H,W) tensor, or
        the string "surface".  Returns (mapping (N,3) int64, weight (N,) float64) on the device."""
        lib = _lib.load()
        if self.intrinsics is not None:   # global intrinsics (fusion_utils.py:38-39)
            intrinsic = self.intrinsics
        k = np.asarray(intrinsic, dtype=np.float64)
        dev = self.device
        W, H = int(self.image_dim[0]), int(self.image_dim[1])
        wvt = torch.as_tensor(world_to_camera, dtype=torch.float32, device=dev).contiguous()
        xyz = torch.as_tensor(coords, dtype=torch.float32, device=dev).contiguous()
        if wvt.shape != (4, 4) or xyz.dim() != 2 or xyz.shape[1] != 3:
            raise RuntimeError("world_to_camera must be (4,4) and coords (N,3)")
        N = xyz.shape[0]
        mapping = torch.empty(N, 3, dtype=torch.int64, device=dev)
        weight = torch.empty(N, dtype=torch.float64, device=dev)
        mode, dptr, zbuf, keep = 0, None, None, None
        if isinstance(depth, str):
            mode = 2
            keep = torch.empty(H, W, dtype=torch.float64, device=dev)
            zbuf = keep.data_ptr()
        elif depth is not None:
            mode = 1
            keep = torch.as_tensor(depth, device=dev).to(torch.float32).contiguous()
            if keep.shape != (H, W):
                raise RuntimeError(f"depth must be (H,W) = ({H},{W}), got {tuple(keep.shape)}")
            dptr = keep.data_ptr()
        intr = (C.c_double * 4)(float(k[0][0]), float(k[1][1]), float(k[0][2]), float(k[1][2]))
        with torch.cuda.device(dev):
            rc = lib.sgs_fusion_compute_mapping(N, xyz.data_ptr(), wvt.data_ptr(), intr, W, H, int(self.cut_bound),
                                                float(self.vis_thres), mode, dptr, zbuf, mapping.data_ptr(),
                                                weight.data_ptr(), _stream(dev))
        _lib.check(rc, "fusion compute_mapping failed")
        return mapping, weight

    def compute_mapping(self, world_to_camera, coords, depth=None, intrinsic=None):
        """The reference's signature and return types (NumPy)."""
        mapping, weight = self.compute_mapping_device(world_to_camera, coords, depth, intrinsic)
        return mapping.cpu().numpy(), weight.cpu().numpy()


def accumulate_features(feat_sum, times, features, mapping, channel_last=False):
    """fusion.py:139-147 for one view, in place on the device: visible points (mapping[:,2] != 0) add the
    feature vector of their pixel to `feat_sum` (N,C) and 1 to `times` (N,).  `features` is the view's 2-D
    feature map, (C,H,W) as the reference has it (transposed to channel-last here, once) or (H,W,C) with
    channel_last=True; `mapping` is compute_mapping_device's (N,3) int64 tensor."""
    lib = _lib.load()
    dev = feat_sum.device
    if feat_sum.dtype != torch.float32 or times.dtype != torch.float32 or not feat_sum.is_contiguous():
        raise RuntimeError("feat_sum (N,C) and times (N,) must be contiguous float32 tensors")
    f = features.to(device=dev, dtype=torch.float32)
    f = f.contiguous() if channel_last else f.permute(1, 2, 0).contiguous()
    H, W, Cn = f.shape
    N = feat_sum.shape[0]
    if feat_sum.shape[1] != Cn or mapping.shape != (N, 3) or mapping.dtype != torch.int64:
        raise RuntimeError("shape mismatch between feat_sum, features and mapping")
    t = times.view(-1)
    with torch.cuda.device(dev):
        rc = lib.sgs_fusion_accumulate(N, Cn, f.data_ptr(), W, H, mapping.contiguous().data_ptr(), feat_sum.data_ptr(),
                                       t.data_ptr(), _stream(dev))
    _lib.check(rc, "fusion accumulate failed")
    return feat_sum, times

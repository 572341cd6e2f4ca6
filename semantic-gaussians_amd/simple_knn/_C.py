"""Stand-in for `simple_knn._C` (SK/ext.cpp:15-17):

    from simple_knn._C import distCUDA2        # model/gaussian_model.py:18
    dist2 = distCUDA2(points_cuda_float32)     # (P,3) -> (P,)
"""
from sgs_hip.raster import dist2 as distCUDA2

__all__ = ["distCUDA2"]

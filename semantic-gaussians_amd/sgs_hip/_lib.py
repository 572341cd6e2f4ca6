"""ctypes binding of libsgs_hip.so (C-ABI declared in include/sgs_raster.h).

The product path has NO fallback: if the HIP library is missing or does not export the
expected ABI the import fails loudly.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(os.path.dirname(_HERE), "csrc")
LIB_PATH = os.path.join(_HERE, "libsgs_hip.so")
ABI_VERSION = 1

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)

SGS_EINVAL, SGS_EHIP, SGS_EALLOC, SGS_ETRAP = -1, -2, -3, -4


class GeometryLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in (
        "depths", "clamped", "radii", "means2D", "cov3D", "conic_opacity", "rgb",
        "tiles_touched", "point_offsets", "total")]


class BinningLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in (
        "keys_unsorted", "vals_unsorted", "keys_sorted", "point_list", "total")]


class ImageLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ("accum_alpha", "n_contrib", "ranges", "total")]


# every symbol include/sgs_raster.h declares
EXPORTS = (
    "sgs_abi_version", "sgs_last_error", "sgs_rasterize_forward", "sgs_rasterize_backward",
    "sgs_mark_visible", "sgs_knn_mean_dist2", "sgs_geometry_layout_of", "sgs_binning_layout_of",
    "sgs_image_layout_of", "sgs_sort_bits", "sgs_debug_expf", "sgs_debug_sorted_keys", "sgs_set_blend_variant",
    "sgs_set_stage_timing", "sgs_get_stage_ms", "sgs_set_binning_mode", "sgs_set_backward_mode", "sgs_build_flags", "sgs_fusion_compute_mapping", "sgs_fusion_accumulate", "sgs_composite_over",
    "sgs_stream_set_option", "sgs_stream_get_stat", "sgs_stream_release", "sgs_debug_set_sweep_trace",
    "sgs_forward_result", "sgs_debug_depth_sort",
    "sgs_device_cu_count", "sgs_stream_create_cu_range", "sgs_stream_destroy", "sgs_stream_set_front", "sgs_x16_cu_ownership",
)

# sgs_stream_set_option / sgs_stream_get_stat selectors (include/sgs_raster.h)
OPT_BLEND_VARIANT, OPT_BINNING_MODE, OPT_BACKWARD_MODE, OPT_STAGE_TIMING, OPT_OUT_PITCH, OPT_DEFER_COUNT = 0, 1, 2, 3, 4, 5
OPT_OUT_BANDS = 8
OPT_BWD_CLEARS_DCOLOR = 6
OPT_NORM_PLANE = 7
STAT_ARENA_SLOTS, STAT_FWD_OVERFLOWS, STAT_BWD_OVERFLOWS, STAT_FORWARDS, STAT_DEFERRED_FORWARDS, STAT_DEFERRED_RETRIES, STAT_BWD_POOL_FALLBACKS, STAT_TILE_ORDER_ALLOC_FAILURES = 0, 1, 2, 3, 4, 5, 6, 7
ERETRY, ENOTREADY = -5, -6

_lib = None


def build(force=False):
    """Compile libsgs_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    args = ["make", "-s", "-C", _CSRC, "-j8"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    # torch FIRST: its wheel ships its own libamdhip64.so (SONAME libamdhip64.so.7, requested by torch's libraries under the unversioned file
    # name).  Loaded after it, libsgs_hip.so's DT_NEEDED libamdhip64.so.7 resolves to that same runtime; loaded BEFORE it, the loader would
    # take /opt/rocm's copy for this library and torch would still bring its own -- two HIP runtimes in one process, and every stream or
    # pointer torch hands to this library would belong to the other one ("no ROCm-capable device is detected" from the first launch).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            f"`make -C {_CSRC}` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise ImportError(f"{LIB_PATH} does not export {name}")
    lib.sgs_abi_version.restype = C.c_int
    if lib.sgs_abi_version() != ABI_VERSION:
        raise ImportError("libsgs_hip.so ABI version mismatch; rebuild the extension")
    lib.sgs_last_error.restype = C.c_char_p
    p, f, i = C.c_void_p, C.c_float, C.c_int
    lib.sgs_rasterize_forward.restype = i
    lib.sgs_rasterize_forward.argtypes = [
        ALLOC_FN, p, ALLOC_FN, p, ALLOC_FN, p, i, i, i, p, i, i, p, p, p, p, p, f, p, p, p, p, p,
        f, f, i, i, p, p, p, i, p]
    lib.sgs_rasterize_backward.restype = i
    lib.sgs_rasterize_backward.argtypes = [
        i, i, i, i, p, i, i, p, p, p, p, f, p, p, p, p, p, f, f, p, p, p, p, p, i,
        p, p, p, p, p, p, p, p, p, i, p]
    lib.sgs_mark_visible.restype = i
    lib.sgs_mark_visible.argtypes = [i, p, p, p, p, p]
    lib.sgs_knn_mean_dist2.restype = i
    lib.sgs_knn_mean_dist2.argtypes = [i, p, p, ALLOC_FN, p, p]
    lib.sgs_geometry_layout_of.restype = i
    lib.sgs_geometry_layout_of.argtypes = [i, C.POINTER(GeometryLayout)]
    lib.sgs_binning_layout_of.restype = i
    lib.sgs_binning_layout_of.argtypes = [i, C.POINTER(BinningLayout)]
    lib.sgs_image_layout_of.restype = i
    lib.sgs_image_layout_of.argtypes = [i, i, C.POINTER(ImageLayout)]
    lib.sgs_sort_bits.restype = i
    lib.sgs_sort_bits.argtypes = [i, i]
    lib.sgs_debug_sorted_keys.restype = i
    lib.sgs_debug_sorted_keys.argtypes = [i, i, i, i, p, p, p, p]
    lib.sgs_debug_expf.restype = i
    lib.sgs_debug_expf.argtypes = [i, p, p, p]
    lib.sgs_set_blend_variant.restype = i
    lib.sgs_set_blend_variant.argtypes = [i]
    lib.sgs_set_stage_timing.restype = i
    lib.sgs_set_stage_timing.argtypes = [i]
    lib.sgs_set_binning_mode.restype = i
    lib.sgs_set_binning_mode.argtypes = [i]
    lib.sgs_set_backward_mode.restype = i
    lib.sgs_set_backward_mode.argtypes = [i]
    lib.sgs_build_flags.restype = i
    lib.sgs_build_flags.argtypes = []
    lib.sgs_fusion_compute_mapping.restype = i
    lib.sgs_fusion_compute_mapping.argtypes = [i, p, p, C.POINTER(C.c_double), i, i, i, C.c_double, i, p, p, p, p, p]
    lib.sgs_composite_over.restype = i
    lib.sgs_composite_over.argtypes = [i, p, p, p, p, p, i, i, i, p]
    lib.sgs_fusion_accumulate.restype = i
    lib.sgs_fusion_accumulate.argtypes = [i, i, p, i, i, p, p, p, p]
    lib.sgs_debug_depth_sort.restype = C.c_longlong
    lib.sgs_debug_depth_sort.argtypes = [i, p, p, p, p]
    lib.sgs_forward_result.restype = i
    lib.sgs_forward_result.argtypes = [p, i, C.POINTER(i)]
    lib.sgs_debug_set_sweep_trace.restype = None
    lib.sgs_debug_set_sweep_trace.argtypes = [p]
    lib.sgs_get_stage_ms.restype = i
    lib.sgs_get_stage_ms.argtypes = [C.POINTER(C.c_float)]
    lib.sgs_stream_set_option.restype = i
    lib.sgs_stream_set_option.argtypes = [p, i, i]
    lib.sgs_stream_get_stat.restype = i
    lib.sgs_stream_get_stat.argtypes = [p, i, C.POINTER(C.c_uint64)]
    lib.sgs_stream_release.restype = i
    lib.sgs_stream_release.argtypes = [p]
    lib.sgs_device_cu_count.restype = i
    lib.sgs_device_cu_count.argtypes = []
    lib.sgs_stream_create_cu_range.restype = i
    lib.sgs_stream_create_cu_range.argtypes = [i, i, C.POINTER(p)]
    lib.sgs_stream_destroy.restype = i
    lib.sgs_stream_destroy.argtypes = [p]
    lib.sgs_stream_set_front.restype = i
    lib.sgs_stream_set_front.argtypes = [p, p]
    lib.sgs_x16_cu_ownership.restype = i
    lib.sgs_x16_cu_ownership.argtypes = []
    _lib = lib
    return lib


def last_error():
    return load().sgs_last_error().decode("utf-8", "replace")


def check(rc, what):
    """Negative return code -> the exception type the reference raises for that failure."""
    if rc >= 0:
        return rc
    msg = last_error() or what
    raise RuntimeError(msg)

"""Round 6: a frame split over two compute-unit partitions (include/sgs_raster.h "compute-unit partitions"; sgs_hip.raster.PartitionedStreams) --
the front end on a stream confined to a few CUs, the blend on a stream confined to the rest -- must give the frame of one ordinary stream,
bit for bit, on every host pattern; and the x16-MFMA kernels must report that they own their compute unit on this device."""
import numpy as np
import pytest
import torch

from helpers import small_scene, oracle_forward

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _render(s, c, C, H, W, fn, pool=None, want_depth=False):
    e = torch.Tensor([])
    return fn(s.bg, s.means3D, s.features, s.opacities, s.scales, s.rotations, 1.0, e, c.world_view_transform, c.full_proj_transform,
              c.tanfovx, c.tanfovy, H, W, e, 0, c.camera_center, False, False, C, want_depth, pool=pool)


def test_x16_kernels_own_their_compute_unit_on_this_device():
    from sgs_hip import raster
    assert raster.x16_cu_ownership() == 3   # (anything else: the x8 sweep / fp32-product backward would be running in their place, silently slower)


@pytest.mark.parametrize("front_cus", [16, 48])
@pytest.mark.parametrize("C", [3, 128, 256])
def test_partitioned_frame_equals_the_one_stream_frame(orc, front_cus, C):
    from sgs_hip import raster
    W, H = 336, 208
    scene, cam = small_scene(P=6000, C=C, W=W, H=H, fx=300.0, seed=21 + C)
    s, c = scene.to(DEV), cam.to(DEV)
    ref = _render(s, c, C, H, W, raster.rasterize_forward, want_depth=(C == 3))
    fw = oracle_forward(orc, scene, cam, want_depth=(C == 3))
    assert ref[0] == fw["num_rendered"]
    ps = raster.PartitionedStreams(DEV, front_cus, 2)
    try:
        assert ps.cu_count == torch.cuda.get_device_properties(0).multi_processor_count
        pools = [raster.ScratchPool(), raster.ScratchPool()]
        for st in ps.streams:
            st.wait_stream(torch.cuda.current_stream())
        for rep in range(3):
            outs = []
            for i, st in enumerate(ps.streams):   # two frames in flight, each split over its two streams
                with torch.cuda.stream(st):
                    fn = (raster.rasterize_forward, raster.rasterize_forward_inference)[(rep + i) % 2]
                    outs.append(_render(s, c, C, H, W, fn, pool=pools[i], want_depth=(C == 3)))
            torch.cuda.synchronize()
            for o in outs:
                assert o[0] == ref[0]
                assert torch.equal(o[1], ref[1]) and torch.equal(o[2], ref[2])
                if C == 3:
                    assert torch.equal(o[6], ref[6])
                iv, rv = raster.image_views(o[5], W, H), raster.image_views(ref[5], W, H)
                for k in ("final_T", "n_contrib", "ranges"):
                    assert torch.equal(iv[k], rv[k]), k
        # the deferred-count pattern: nothing waits for the GPU between the frames
        hs = []
        for i, st in enumerate(ps.streams):
            with torch.cuda.stream(st):
                hs.append(_render(s, c, C, H, W, raster.rasterize_forward_deferred, pool=pools[i], want_depth=(C == 3)))
        for h in hs:
            o = h.result()
            assert o[0] == ref[0]
        torch.cuda.synchronize()
        for h in hs:
            assert torch.equal(h.result()[1], ref[1])
    finally:
        ps.close()


def test_partitioned_forward_feeds_the_backward(orc):
    """A differentiated frame on a partitioned stream: the backward (one stream, the blend partition) reads the buffers both streams wrote."""
    from sgs_hip import raster
    C, W, H = 64, 208, 160
    scene, cam = small_scene(P=3000, C=C, W=W, H=H, fx=200.0, seed=77)
    s, c = scene.to(DEV), cam.to(DEV)
    e = torch.Tensor([])
    dL = torch.randn(C, H, W, device=DEV, generator=torch.Generator(DEV).manual_seed(3))

    def fwd_bwd():
        n, col, rad, g, b, i, _ = _render(s, c, C, H, W, raster.rasterize_forward)
        return col, raster.rasterize_backward(s.bg, s.means3D, rad, s.features, s.scales, s.rotations, 1.0, e, c.world_view_transform,
                                              c.full_proj_transform, c.tanfovx, c.tanfovy, dL, e, 0, c.camera_center, g, n, b, i, False)
    col0, g0 = fwd_bwd()
    ps = raster.PartitionedStreams(DEV, 32, 1)
    try:
        ps.streams[0].wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(ps.streams[0]):
            col1, g1 = fwd_bwd()
        torch.cuda.synchronize()
        assert torch.equal(col0, col1)
        for a, b in zip(g0, g1):   # (atomics: the order of the sums is not fixed)
            if a.numel() == 0:
                continue
            scale = float(a.abs().max()) + 1e-30
            assert float((a - b).abs().max()) <= 1e-5 * scale
    finally:
        ps.close()


@pytest.mark.parametrize("cus", [4, 32])
@pytest.mark.parametrize("kind", ["random", "depth"])
def test_chained_sort_passes_on_a_few_compute_units(cus, kind):
    """Round 6: the depth sort's passes 1 .. 3 wait, inside the kernel, for the count rows of the tiles in front of theirs (csrc/depth_sort.hip CHAIN).
    A workgroup takes its tile from a ticket, so what it waits for always belongs to a workgroup that is already running -- also when only a handful
    of the 245 (1 M keys) / 733 (3 M keys) workgroups are resident at a time: here the sort runs on a stream confined to 4 / 32 compute units."""
    import ctypes as C
    from sgs_hip import _lib
    lib = _lib.load()
    st = C.c_void_p()
    assert lib.sgs_stream_create_cu_range(0, cus, C.byref(st)) == 0
    try:
        for P in (1_000_000, 3_000_001):
            g = torch.Generator(device=DEV).manual_seed(P + cus)
            if kind == "random":
                keys = torch.randint(0, 2 ** 32, (P,), device=DEV, generator=g, dtype=torch.int64)
            else:
                keys = (0.2 + 60.0 * torch.rand(P, device=DEV, generator=g)).view(torch.int32).to(torch.int64)
                keys[torch.rand(P, device=DEV, generator=g) < 0.1] = 0xFFFFFFFF
            want = torch.sort(keys, stable=True).indices.to(torch.int32)
            k32 = torch.where(keys >= 2 ** 31, keys - 2 ** 32, keys).to(torch.int32)   # same bits as uint32
            need = lib.sgs_debug_depth_sort(P, None, None, None, None)
            scratch = torch.empty(need, dtype=torch.uint8, device=DEV)
            perm = torch.full((P,), -1, dtype=torch.int32, device=DEV)
            torch.cuda.synchronize()
            for _ in range(2):
                assert lib.sgs_debug_depth_sort(P, k32.data_ptr(), perm.data_ptr(), scratch.data_ptr(), st) == 0
                torch.cuda.synchronize()
                assert torch.equal(perm, want)
    finally:
        lib.sgs_stream_destroy(st)


def test_cu_range_arguments():
    from sgs_hip import raster, _lib
    import ctypes as C
    lib = _lib.load()
    n = lib.sgs_device_cu_count()
    assert n == torch.cuda.get_device_properties(0).multi_processor_count
    st = C.c_void_p()
    assert lib.sgs_stream_create_cu_range(0, n + 1, C.byref(st)) == _lib.SGS_EINVAL
    assert lib.sgs_stream_create_cu_range(-1, 4, C.byref(st)) == _lib.SGS_EINVAL
    assert lib.sgs_stream_create_cu_range(0, 0, C.byref(st)) == _lib.SGS_EINVAL
    assert lib.sgs_stream_create_cu_range(8, 8, C.byref(st)) == 0 and st.value
    assert lib.sgs_stream_destroy(st) == 0
    with pytest.raises(ValueError):
        raster.PartitionedStreams(DEV, 0, 1)

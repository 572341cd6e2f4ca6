"""Gaussian sharding on real devices (BASELINE config 5's exchange step): the composite kernel on one GPU, and -- when
the box has at least two -- dist.render_gaussian_sharded over RCCL with HIP partials under torch.distributed.run."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


@pytest.mark.parametrize("S,C,h,W", [(1, 5, 16, 64), (3, 37, 24, 100), (8, 256, 32, 1296), (16, 4, 7, 12)])
def test_composite_over_kernel_equals_the_torch_chain(S, C, h, W):
    from sgs_hip import dist as sdist
    g = torch.Generator(device=DEV).manual_seed(S * 1000 + C)
    partials = [(torch.randn(C, h, W, device=DEV, generator=g), torch.rand(h, W, device=DEV, generator=g)) for _ in range(S)]
    bg = torch.randn(C, device=DEV, generator=g)
    for use_bg in (True, False):
        out, t = sdist.composite_over(partials, bg if use_bg else None)
        ref, tr = partials[0][0].clone(), partials[0][1].clone()
        for A, T in partials[1:]:
            ref += tr.unsqueeze(0) * A
            tr = tr * T
        if use_bg:
            ref += bg.reshape(-1, 1, 1) * tr.unsqueeze(0)
        assert torch.equal(out, ref) and torch.equal(t, tr)


@pytest.mark.parametrize("P,C,W,H,fx,nb,dense", [(4000, 128, 208, 160, 170.0, 2, False), (3000, 256, 205, 117, 170.0, 3, False),
                                                  (3000, 128, 100, 70, 90.0, 8, False), (40000, 128, 400, 96, 300.0, 4, True)])
def test_band_major_partials_equal_the_plain_partial(P, C, W, H, fx, nb, dense):
    """SGS_OPT_OUT_BANDS (raster.render_partial(..., bands=n)): the kernels write the shard's feature map band-major -- band b of
    dist.band_rows as one contiguous (C, rows, W) block -- so that the image-partitioned exchange of a Gaussian-sharded view sends every
    band as it lies.  Each band must hold exactly the bits of the plain (C,H,W) partial's rows: the sweep, and (dense scene: first frame
    on a cold stream) the gated single-kernel fallback that renders a frame whose work list overflowed; more bands than tile rows give
    empty bands; widths that are not a multiple of 16."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import small_scene
    from sgs_hip import raster, dist as sdist
    scene, cam = small_scene(P=P, C=C, W=W, H=H, fx=fx, seed=P + nb)
    if dense:
        scene = scene._replace(scales=scene.scales * 3.0, opacities=scene.opacities * 0.05)
    s, c = scene.to(DEV), cam.to(DEV)
    args = (s.means3D, s.features, s.opacities, s.scales, s.rotations, c.world_view_transform, c.full_proj_transform, c.tanfovx, c.tanfovy,
            H, W, c.camera_center)
    st = torch.cuda.Stream(DEV)   # a cold stream: the dense scene's first frame overflows its work list (the fallback renders it)
    with torch.cuda.stream(st):
        for rep in range(3):
            bands, tb, rb = raster.render_partial(*args, bands=nb)
            plain, tp, rp = raster.render_partial(*args)
            torch.cuda.synchronize()
            assert isinstance(bands, list) and len(bands) == nb
            assert torch.equal(tb, tp) and torch.equal(rb, rp)
            for b in range(nb):
                lo, hi = sdist.band_rows(H, b, nb)
                assert bands[b].is_contiguous() and tuple(bands[b].shape) == (C, hi - lo, W), (b, bands[b].shape)
                if dense and rep == 0:   # this frame was rendered by the fallback's fp32 chain, the plain one behind it by the sweep's six products
                    assert float((bands[b] - plain[:, lo:hi]).abs().max()) <= 2e-5 * float(plain.abs().max()), (rep, b)
                else:
                    assert torch.equal(bands[b], plain[:, lo:hi]), (rep, b)
            assert bands[0].untyped_storage().data_ptr() == bands[-1].untyped_storage().data_ptr()   # views of ONE buffer
        if dense:
            assert raster.stream_stat(__import__("sgs_hip._lib", fromlist=["x"]).STAT_FWD_OVERFLOWS) >= 1
        raster.release_stream()


@pytest.mark.parametrize("variant", [15, 14])
def test_banded_partial_under_a_variant_without_band_output(variant):
    """render_partial(bands = n) under a blend variant whose sweep has no band-major output (the exact fp32 sweep, round 2's two-term
    sweep) used to raise (the library answers SGS_OPT_OUT_BANDS with SGS_EINVAL there); it now renders row-major and cuts the same bands."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import small_scene
    from sgs_hip import raster, dist as sdist
    C, W, H, nb = 128, 208, 160, 3
    scene, cam = small_scene(P=4000, C=C, W=W, H=H, fx=170.0, seed=5)
    s, c = scene.to(DEV), cam.to(DEV)
    args = (s.means3D, s.features, s.opacities, s.scales, s.rotations, c.world_view_transform, c.full_proj_transform, c.tanfovx, c.tanfovy,
            H, W, c.camera_center)
    raster.set_blend_variant(variant)
    try:
        bands, tb, rb = raster.render_partial(*args, bands=nb)
        plain, tp, rp = raster.render_partial(*args)
    finally:
        raster.set_blend_variant(0)
    assert isinstance(bands, list) and len(bands) == nb and torch.equal(tb, tp) and torch.equal(rb, rp)
    for b in range(nb):
        lo, hi = sdist.band_rows(H, b, nb)
        assert bands[b].is_contiguous() and torch.equal(bands[b], plain[:, lo:hi])


def test_band_major_partial_of_an_empty_shard():
    """A depth slab with nothing in front of the camera (num_rendered == 0: the forward takes its no-work-list path) still hands over
    band-major zeros with transmittance one -- the shape the exchange expects from every rank."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import small_scene
    from sgs_hip import raster, dist as sdist
    C, W, H, nb = 128, 120, 100, 3
    scene, cam = small_scene(P=500, C=C, W=W, H=H, fx=100.0, seed=5)
    s, c = scene.to(DEV), cam.to(DEV)
    view_z = (torch.cat([s.means3D, torch.ones_like(s.means3D[:, :1])], 1) @ c.world_view_transform)[:, 2]
    keep = view_z < 0.0   # behind the camera: culled by the frustum test
    means = s.means3D.clone()
    if int(keep.sum()) == 0:   # none there: mirror every Gaussian through the camera centre
        means = 2.0 * c.camera_center.reshape(1, 3) - means
        keep = torch.ones_like(keep)
    bands, tb, rb = raster.render_partial(means[keep], s.features[keep], s.opacities[keep], s.scales[keep], s.rotations[keep],
                                          c.world_view_transform, c.full_proj_transform, c.tanfovx, c.tanfovy, H, W, c.camera_center, bands=nb)
    torch.cuda.synchronize()
    assert len(bands) == nb and float(tb.min()) == 1.0 and float(tb.max()) == 1.0
    for b in range(nb):
        lo, hi = sdist.band_rows(H, b, nb)
        assert tuple(bands[b].shape) == (C, hi - lo, W) and bands[b].is_contiguous() and float(bands[b].abs().max()) == 0.0
    raster.release_stream()


def test_gaussian_sharded_render_over_rccl_two_gpus():
    """torchrun --nproc 2: each rank renders one depth slab with the HIP rasteriser (raster.render_partial), the bands
    travel as grouped point-to-point sends / receives over RCCL, every rank composites its band with the HIP kernel;
    rank 0 compares the gathered map with its own single render."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(29700 + os.getpid() % 200),
                        os.path.join(ROOT, "tools", "run_gaussian_sharded.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "GAUSSIAN_SHARDED_OK" in r.stdout


def test_gaussian_sharded_render_two_ranks_on_one_device_over_gloo():
    """The same script as the RCCL test with both ranks on cuda:0 and gloo as the transport (test hook): HIP partials of two
    depth slabs, the band exchange as grouped point-to-point operations on device tensors, the HIP composite, the gathered
    map against rank 0's single render -- everything but RCCL itself, on a one-GPU box."""
    env = dict(os.environ, SGS_TEST_ONE_DEVICE="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(29300 + os.getpid() % 200),
                        os.path.join(ROOT, "tools", "run_gaussian_sharded.py"), "60000", "128"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "GAUSSIAN_SHARDED_OK" in r.stdout


def test_bench_two_ranks_control_flow_on_one_device():
    """bench.py --gpus 2 end to end on ONE device (test hook SGS_BENCH_TEST_ONE_DEVICE: both ranks on cuda:0, gloo for the
    control collectives): rank spawn, rendezvous on 127.0.0.1, barriers, max-over-ranks timing, the aggregate value, one
    JSON line from rank 0 -- and the watchdog that prints the line when configs 4 / 5 do not finish in time."""
    import json
    env = dict(os.environ, SGS_BENCH_TEST_ONE_DEVICE="1", SGS_BENCH_EXTRA_TIMEOUT_S="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["config"]["rccl"]["ranks_seen"] == 2
    assert d["integrity"]["num_rendered_mismatches_vs_serial"] == 0
    assert "error" in d["multi_gpu_configs"]      # the 1 s watchdog fired; everything else is complete
    assert abs(d["value"] - 2 * d["config"]["views_per_step_per_gpu"] * 968 * 1296 * 512 / (d["ms_per_step"] * 1e-3) / 1e9) < 1e-6 * d["value"]

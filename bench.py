#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X N-channel Gaussian-splat rasteriser.

Metric (BASELINE.json): Gpixel*channels/s of the FORWARD feature render,
    1M synthetic Gaussians, C = 512, 968x1296   ("cfg3")
  = H*W*C / t_fwd / 1e9, t_fwd = the whole rasterize_gaussians forward (preprocess -> binning -> blend),
    inputs already resident in HBM, called through the C-ABI.

What one JSON line carries (rank 0):
  value / ms_per_step   the contract's number: K timed steps between barriers, a step = `--views` (default 4) views of
                        the scene in flight on as many HIP streams, every forward returning its num_rendered to the host
                        as the reference's does (round 5: through raster.rasterize_forward_inference -- the frame is enqueued
                        in full before the host waits for the count, what the drop-in module does under torch.no_grad();
                        `classic_count` = the reference's mid-frame wait, rounds 1-4's headline) and a different camera every step; DEFAULT
                        arithmetic of the C >= 128 blend: "f32-equivalent" -- features and weights split exactly into
                        three bf16 terms, six MFMA products, fp32 accumulate (as accurate as the reference's fp32 chain
                        against the exact composite: tests/test_configs_gpu.py) -- `dtype` says so.  The default K keeps
                        >= 2 s of continuous GPU work in the timed region;
  api_path              what a user of the reference gets: channel_rasterization.GaussianRasterizer called exactly as
                        model/renderer.py:169-185,228 does (nn.Parameter inputs, torch.no_grad(), debug=True, one view;
                        the host waits for num_rendered before the call returns, after the whole frame was enqueued),
                        device ms per forward (hipEvents, median);
  single_view           one view in flight through the internal entry point (SURVEY.md 8(d)'s t_fwd): value, ms_median -- measured right
                        behind the timed region (steady state; round 6); single_view_cold = the same leg at the top of the process, where
                        rounds 1-5 measured it (the sweep reads 5-6 % slower there);
  single_view_inference the same through raster.rasterize_forward_inference (the frame enqueued in full before the host waits for
                        the count: what the drop-in module does under no_grad);
  deferred_count        the inference-only mode without the host read-back (SGS_OPT_DEFER_COUNT), V views in flight,
                        with the number of frames that had to be rendered twice;
  exact_f32             the bit-exact fp32-MFMA arithmetic (SGS_BLEND_EXACT=1 / variant 15);
  two_term              round 2's default arithmetic (variant 14: two bf16 terms, three products, 3 * 2^-16 per term --
                        NOT fp32-class; kept selectable), for continuity with BENCH_r02;
  backward              cfg3 is "forward+backward": forward+backward device ms of the same scene, the backward alone
                        and its algorithmic-bytes fraction of the HBM roofline (N = 1 only);
  roofline              the forward blend against the HBM roofline: achieved = SURVEY 8(d)'s algorithmic bytes of
                        the blend / its kernels' live hipEvent durations (one view in flight); plus the secondary
                        ceilings (fp32 FMA, bf16 MFMA) the same work is priced against;
  semantic_consumer     SURVEY 8(f) N1: per-view text similarities three ways (the reference's flow on this rasteriser, the
                        normalised values via the norm-plane epilogue, the unnormalised logits), N = 1 only;
  multi_gpu_configs     BASELINE configs 4 (views sharded) and 5 (Gaussians sharded, RCCL band exchange) on this job's
                        ranks: with --extra-configs, and by default when N > 1 (after the line is complete, under a
                        watchdog);
  cpu_baseline          kind "port" (a restatement, not the reference itself, which cannot run here), implementation pure
                        PyTorch: the CPU splat the north star names (oracle/torch_splat.py): cfg1 in
                        full, cfg3 on a tile sample extrapolated by the tiles' list work; cpu_baseline_port: the
                        C/OpenMP oracle.

--gpus N: one process per GPU.  Under torch.distributed.run the ranks are given; a plain `python bench.py --gpus N`
spawns its own N ranks (127.0.0.1 rendezvous).  Views shard across ranks with no data-path collective (scene
replicated), so per-GPU work is fixed ("scaling": "weak") and `value` is the whole-job aggregate; RCCL carries
only the barriers and the max-over-ranks of the time.
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "semantic-gaussians_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

# keep the caching allocator from splitting the multi-GB output / scratch blocks (otherwise it
# needs tens of frames of hipMalloc before it settles)
os.environ.setdefault("PYTORCH_HIP_ALLOC_CONF", "max_split_size_mb:256")
os.environ.setdefault("PYTORCH_CUDA_ALLOC_CONF", os.environ["PYTORCH_HIP_ALLOC_CONF"])

import numpy as np  # noqa: E402
import torch  # noqa: E402

EXTRA_TIMEOUT_S = float(os.environ.get("SGS_BENCH_EXTRA_TIMEOUT_S", "240"))   # configs 4 / 5 (bench.py --extra-configs, on by default when --gpus > 1)
HBM_PEAK = 8.0e12        # MI355X spec (MI355X_MICROARCH.md "Chip-level parameters")
FP32_FMA_PEAK = 157.3e12  # fp32 vector (= fp32-input MFMA) peak, same table
BF16_MFMA_PEAK = 2.5e15   # dense bf16 MFMA peak
STAGES = ["preprocess", "scan_readback", "duplicate", "sort", "ranges", "blend_weights", "blend_accum"]
EXACT = 15               # blend variant: fp32-input MFMA accumulate, bit-identical to the contract
TWO_TERM = 14            # blend variant: round 2's two-term split (three products)
NCAM = 8                 # cameras each view slot cycles through (a different view every step)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def view_camera(n, W, H, fx):
    """View number n: the generator's camera (at the origin, looking down +z at the frustum-shaped slab) yawed and shifted
    by a FEW MILLIRADIANS / millimetres, 16 distinct poses: every view sees the whole slab and carries the workload
    BASELINE's config quotes (num_rendered within 1 % of the centred camera's 16.5 M) -- the frames differ, the work does
    not.  (Until the end of round 3 the offsets grew with n without bound: views 8 .. 31 saw 80 % .. 25 % of the
    Gaussians, views >= 64 none -- a longer camera cycle or more slots would have rendered cheaper frames.)"""
    import math
    from sgs_hip.camera import make_camera, focal2fov
    m = (n % 16) - 7.5
    a = 0.0008 * m
    R = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    T = np.array([0.0010 * m, -0.0005 * m, 0.0])
    return make_camera(R, T, focal2fov(fx, W), focal2fov(fx, H), W, H)


def pin_to_gpu_numa(dev):
    """Pin this rank's host threads to the cores of its GPU's NUMA node (eight Python processes on a 256-thread host
    migrate, and the >= 6x target at 8 GPUs guards against exactly that kind of host-side serialisation).  The node comes from the
    device's PCI address in sysfs (local_cpulist of the function the runtime reports for `dev`); SGS_BENCH_NO_PIN=1 switches it off.
    -> a dict for the bench line (what was done, or why not); never raises."""
    info = {"pinned": False}
    try:
        if os.environ.get("SGS_BENCH_NO_PIN", "0") == "1":
            info["reason"] = "SGS_BENCH_NO_PIN=1"
            return info
        pr = torch.cuda.get_device_properties(dev)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        base = f"/sys/bus/pci/devices/{bdf}"
        info["pci"] = bdf
        with open(base + "/numa_node") as f:
            info["numa_node"] = int(f.read().strip())
        with open(base + "/local_cpulist") as f:
            cpulist = f.read().strip()
        cpus = set()
        for part_ in cpulist.split(","):
            if "-" in part_:
                a, b = part_.split("-")
                cpus.update(range(int(a), int(b) + 1))
            elif part_:
                cpus.add(int(part_))
        allowed = os.sched_getaffinity(0)
        cpus &= allowed
        if not cpus or cpus == allowed:
            info["reason"] = "the device's local cpulist is the whole allowed set (one NUMA node, or no locality information)"
            info["cpus"] = len(allowed)
            return info
        os.sched_setaffinity(0, cpus)
        info.update(pinned=True, cpus=len(cpus), cpulist=cpulist)
    except Exception as ex:   # noqa: BLE001
        info["reason"] = f"{type(ex).__name__}: {ex}"
    return info


def self_spawn(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (rank 0's stdout is ours)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, p.wait())
    sys.exit(rc)


def cpu_baseline_port(scene, cam, C, W, H, n_eff, budget_s=8.0):
    """The C/OpenMP oracle on the host cores: preprocess + binning in full, blend on a sample of whole tile rows
    from the middle of the frame, extrapolated by the tiles' list work (sum n_t_eff of the sample vs the frame)."""
    from oracle import oracle as orc
    nthreads = os.cpu_count() or 1
    t0 = time.time()
    pre = orc.preprocess(scene.means3D.numpy(), scene.opacities.numpy(),
                         cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(),
                         cam.camera_center.numpy(), W, H, cam.tanfovx, cam.tanfovy,
                         scales=scene.scales.numpy(), rotations=scene.rotations.numpy(),
                         colors_precomp=np.zeros((1, 1), np.float32))
    binn = orc.binning(pre, W, H)
    t_front = time.time() - t0
    gx, gy = orc.tile_grid(W, H)
    ntiles = gx * gy
    feats = scene.features.numpy()
    lo = (gy // 2) * gx
    n_sample = min(ntiles - lo, max(gx, 2 * nthreads))
    t_blend, done = 0.0, 0
    while True:
        t0 = time.time()
        orc.blend_forward(pre, binn, feats, scene.bg.numpy(), W, H, tile_lo=lo + done, tile_hi=lo + done + n_sample)
        t_blend += time.time() - t0
        done += n_sample
        if t_blend > budget_s or lo + done + n_sample > ntiles:
            break
    work = float(n_eff[lo:lo + done].sum()) / max(1.0, float(n_eff.sum()))
    t_frame = t_front + t_blend / work
    return dict(value=H * W * C / t_frame / 1e9, unit="Gpixel*channels/s", cores=nthreads, kind="port",
                sample=(f"oracle (C port, OpenMP x{nthreads}): preprocess+binning in full ({t_front:.2f} s) + blend on "
                        f"{done} of {ntiles} tiles ({t_blend:.2f} s) extrapolated by sum n_t_eff "
                        f"({work * 100:.1f} % of the frame's list work) -> {t_frame:.1f} s/frame"))


def cpu_baseline_pytorch(scene, cam, C, W, H, n_eff, budget_s=10.0):
    """The pure-PyTorch CPU splat (oracle/torch_splat.py), all host cores: cfg1 timed in full; this workload's
    preprocess + binning in full and its blend on a tile sample, extrapolated by the tiles' list work."""
    from oracle import torch_splat
    from sgs_hip.synthetic import make_config
    ncores = os.cpu_count() or 1
    # thread count: os.cpu_count() is what the north star names, but torch's intra-op pool can be far slower with every
    # hardware thread of a large host than with a few dozen (measured here: 256 threads were 200x slower than 8 on the
    # cfg1 frame).  Calibrate on one chunk of the blend's own arithmetic and use the fastest; all timings are reported.
    g = torch.Generator().manual_seed(0)
    a_ = torch.rand(64, 256, 128, generator=g)
    f_ = torch.rand(64, 128, max(8, min(C, 64)), generator=g)
    calib = {}
    for n in sorted({ncores, 64, 32, 16, 8}):
        if n > ncores:
            continue
        torch.set_num_threads(n)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            w_ = torch.cumprod(1.0 - torch.exp(-a_), 2) * a_
            torch.bmm(w_, f_)
            best = min(best, time.perf_counter() - t0)
        calib[n] = best
    nthreads = min(calib, key=calib.get)
    torch.set_num_threads(nthreads)
    s1, c1 = make_config("cfg1")
    t1 = {}
    torch_splat.render(s1, c1, 256, 256, timings=t1)     # warm (thread pools, allocator)
    t0 = time.perf_counter()
    torch_splat.render(s1, c1, 256, 256, timings=t1)
    cfg1_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    pre = torch_splat.preprocess(scene.means3D, scene.scales, scene.rotations, scene.opacities,
                                 cam.world_view_transform, cam.full_proj_transform, W, H, cam.tanfovx, cam.tanfovy)
    binn = torch_splat.binning(pre, W, H)
    t_front = time.perf_counter() - t0
    gx, gy = (W + 15) // 16, (H + 15) // 16
    ntiles = gx * gy
    lo = (gy // 2) * gx
    done, t_blend, n = 0, 0.0, 8
    while lo + done < ntiles and t_blend < budget_s:
        n = min(n, ntiles - lo - done)
        t0 = time.perf_counter()
        torch_splat.blend_tiles(pre, binn, scene.features, scene.bg, W, H, tile_ids=range(lo + done, lo + done + n))
        t_blend += time.perf_counter() - t0
        done += n
        n *= 2
    work = float(n_eff[lo:lo + done].sum()) / max(1.0, float(n_eff.sum()))
    t_frame = t_front + t_blend / work
    return dict(value=H * W * C / t_frame / 1e9, unit="Gpixel*channels/s", cores=nthreads, kind="port", implementation="pure PyTorch (oracle/torch_splat.py): the CPU baseline BASELINE.md names",
                host_cpus=ncores, thread_calibration_ms={str(k): round(v * 1e3, 2) for k, v in calib.items()},
                cfg1={"workload": "cfg1: 10k Gaussians, 256x256, C=3, full frame", "seconds": cfg1_s,
                      "value": 256 * 256 * 3 / cfg1_s / 1e9, "unit": "Gpixel*channels/s"},
                sample=(f"pure-PyTorch CPU splat, {torch.get_num_threads()} threads: preprocess+binning in full "
                        f"({t_front:.2f} s) + blend on {done} of {ntiles} tiles ({t_blend:.2f} s) extrapolated by "
                        f"sum n_t_eff ({work * 100:.2f} % of the frame's list work) -> {t_frame:.1f} s/frame"))


def extra_config_legs(rank, world, dev, dist, red_dev=None):
    """BASELINE configs 4 and 5 on this job's ranks.  Returns a dict for the JSON line; never raises."""
    import traceback
    from sgs_hip import raster, dist as sdist
    from sgs_hip.synthetic import CONFIGS, make_scene
    empty = torch.Tensor([])
    out = {}

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(t):
        if world > 1:
            tt = torch.tensor([t], device=red_dev if red_dev is not None else dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        return t
    # ---- config 4: 5M Gaussians x 768 channels, 840x1297, 8 views on a ring sharded round-robin, scene replicated
    try:
        P, C, W, H, fx = CONFIGS["cfg4"]
        log(f"[rank {rank}] config 4: generating the scene")
        scene = make_scene(P, C, W, H, fx, seed=4, features=False)
        s = scene.to(dev)
        g = torch.Generator(device=dev).manual_seed(44)
        feats = torch.empty(P, C, device=dev)
        for i in range(0, P, 1 << 20):
            f = torch.randn(min(P, i + (1 << 20)) - i, C, device=dev, generator=g)
            feats[i:i + f.shape[0]] = f / f.norm(dim=1, keepdim=True)
        del f
        bg = torch.zeros(C, device=dev)
        views = [view_camera(i, W, H, fx).to(dev) for i in range(8)]
        mine = sdist.shard_views(len(views), rank, world)
        pool = raster.ScratchPool()
        raster.OUTPUT_PITCH_ALIGN = 32   # 1297 is not a multiple of 16: rows padded to whole 128-byte lines

        def render4(c):
            return raster.rasterize_forward(bg, s.means3D, feats, s.opacities, s.scales, s.rotations, 1.0, empty, c.world_view_transform,
                                            c.full_proj_transform, c.tanfovx, c.tanfovy, H, W, empty, 0, c.camera_center, False,
                                            False, C, False, pool=pool)[1]
        o = None
        for _ in range(2):   # warm exactly like the timed loop (one result alive while the next allocates)
            for i in mine:
                o = render4(views[i])
        passes = 4
        barrier()
        t0 = time.perf_counter()
        for _ in range(passes):
            for i in mine:
                o = render4(views[i])   # results stay on the GPU that rendered them
        barrier()
        t = max_over_ranks(time.perf_counter() - t0)
        raster.OUTPUT_PITCH_ALIGN = 0
        nv = passes * len(views)
        out["cfg4"] = {"workload": f"cfg4: P={P} x C={C}, {H}x{W}, 8 views round-robin over {world} GPU(s), scene replicated, no collective, "
                                   f"output rows padded to 32 px", "views_per_s": nv / t, "value": nv * H * W * C / t / 1e9,
                       "unit": "Gpixel*channels/s", "ms_per_view_per_gpu": t / (passes * max(1, len(views) // world)) * 1e3, "seconds": t}
        del feats, s, o, pool
        torch.cuda.empty_cache()
    except Exception:   # noqa: BLE001
        raster.OUTPUT_PITCH_ALIGN = 0
        out["cfg4"] = {"error": traceback.format_exc()[-1500:]}
    # ---- config 5: 50M Gaussians x 256 channels Gaussian-sharded: rank r holds view-space depth slab r; (A, T) partials,
    # image-partitioned point-to-point exchange over RCCL, one composite kernel per band
    try:
        P, C, W, H, fx = CONFIGS["cfg5"]
        log(f"[rank {rank}] config 4 done; config 5: generating the scene")
        scene = make_scene(P, C, W, H, fx, seed=5, features=False)   # every rank: the same geometry (2.2 GB on the host)
        order = torch.argsort(scene.means3D[:, 2])                   # camera at the origin looking down +z
        idx = order[rank * P // world:(rank + 1) * P // world]
        sh = [t[idx].to(dev) for t in (scene.means3D, scene.opacities, scene.scales, scene.rotations)]
        del scene, order
        n_loc = idx.numel()
        g = torch.Generator(device=dev).manual_seed(500 + rank)
        feats = torch.empty(n_loc, C, device=dev)
        for i in range(0, n_loc, 1 << 21):
            f = torch.randn(min(n_loc, i + (1 << 21)) - i, C, device=dev, generator=g)
            feats[i:i + f.shape[0]] = f / f.norm(dim=1, keepdim=True)
        del f
        bg = torch.linspace(0.0, 1.0, C, device=dev)
        cam = view_camera(0, W, H, fx).to(dev)
        pool = raster.ScratchPool()

        def partial():
            A, T, _ = raster.render_partial(sh[0], feats, sh[1], sh[2], sh[3], cam.world_view_transform, cam.full_proj_transform,
                                            cam.tanfovx, cam.tanfovy, H, W, cam.camera_center, pool=pool,
                                            bands=world if C % 128 == 0 else 0)   # band-major: every band is sent as it lies
            return A, T
        log(f"[rank {rank}] config 5: slab of {n_loc} Gaussians resident, first frame")
        band = sdist.render_gaussian_sharded(partial, bg, all_gather=False)
        log(f"[rank {rank}] config 5: first frame done")
        frames = 4
        barrier()
        t0 = time.perf_counter()
        for _ in range(frames):
            band = sdist.render_gaussian_sharded(partial, bg, all_gather=False)   # every rank ends up with its band of rows
        barrier()
        t = max_over_ranks(time.perf_counter() - t0)
        lo, hi = sdist.band_rows(H, rank, world)
        out["cfg5"] = {"workload": f"cfg5: P={P} x C={C}, {H}x{W}, Gaussians sharded into {world} view-space depth slab(s) "
                                   f"({n_loc} on this rank), (A, T) partials exchanged as grouped RCCL send / recv by image band, "
                                   f"one composite kernel per band; each rank keeps its band of rows",
                       "frames_per_s": frames / t, "value": frames * H * W * C / t / 1e9, "unit": "Gpixel*channels/s",
                       "ms_per_frame": t / frames * 1e3,
                       "exchange_bytes_sent_per_rank_per_frame": int((C + 1) * 4 * W * (H - (hi - lo))) if world > 1 else 0,
                       "band_rows_rank0": [lo, hi], "finite": bool(torch.isfinite(band).all())}
        del feats, sh, band, pool
        torch.cuda.empty_cache()
    except Exception:   # noqa: BLE001
        out["cfg5"] = {"error": traceback.format_exc()[-1500:]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--points", type=int, default=None, help="override P (debug)")
    ap.add_argument("--channels", type=int, default=None, help="override C (debug)")
    ap.add_argument("--variant", type=int, default=0, help="blend kernel variant of the headline (tuning)")
    ap.add_argument("--views", type=int, default=4,
                    help="views in flight per GPU: a step renders this many views of the scene, one per HIP "
                         "stream, so one view's front-end and host round trip overlap another view's blend")
    ap.add_argument("--classic-count", action="store_true",
                    help="headline with the host waiting for num_rendered in the MIDDLE of every forward (the reference's own host pattern, "
                         "rasterizer_impl.cu:283, and rounds 1-4's headline); default: the inference entry point -- the frame is enqueued in "
                         "full against the stream's capacity guess, then the host waits for the count (what GaussianRasterizer does under no_grad)")
    ap.add_argument("--deferred-count", action="store_true",
                    help="headline with deferred counts (SGS_OPT_DEFER_COUNT, inference only) instead of the reference's "
                         "blocking num_rendered read-back in every forward")
    ap.add_argument("--fixed-camera", action="store_true",
                    help="every step renders the same camera per slot (profiling runs: per-kernel averages of ONE view)")
    ap.add_argument("--extra-configs", action="store_true",
                    help="also time BASELINE configs 4 (views sharded) and 5 (Gaussians sharded, RCCL exchange); on by default "
                         "when --gpus > 1")
    ap.add_argument("--front-cus", type=int, default=int(os.environ.get("SGS_BENCH_FRONT_CUS", "0")),
                    help="compute units set aside for the views' front ends in the headline (raster.PartitionedStreams: every view slot gets a "
                         "blend stream on the other CUs and a front stream on these); 0 (default) = ordinary streams, every kernel anywhere.  Measured "
                         "(profiles/r06_cu_partition.txt): the sweep slows in proportion to the CUs it loses -- 1.02 ms on 256, 1.17 on 224, 1.33 on 208 -- "
                         "which costs more than the front end's interference it removes (465 vs 477 Gpx.ch/s at 32 front CUs, four views in flight)")
    ap.add_argument("--blend-everywhere", action="store_true",
                    help="with --front-cus: only the front ends are confined; the blend streams may use every compute unit")
    ap.add_argument("--stream-priority", choices=("none", "split", "blend", "front"), default=os.environ.get("SGS_BENCH_STREAM_PRIORITY", "none"),
                    help="experiment (profiles/r06_stream_priority.txt): every view slot's front end on a second ordinary stream (split), with the blend "
                         "stream (blend) or the front stream (front) at high priority; none (default) = one stream per slot")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the exact-arithmetic / backward legs")
    args = ap.parse_args()
    global NCAM
    if args.fixed_camera:
        NCAM = 1

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_spawn(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # SGS_BENCH_TEST_ONE_DEVICE=1 (test hook, never set by the driver): every rank on cuda:0 and gloo for the control
    # collectives, so that the N > 1 control flow can be exercised on a single-GPU box (RCCL refuses two ranks per device)
    one_dev = os.environ.get("SGS_BENCH_TEST_ONE_DEVICE", "0") == "1"
    dev = torch.device("cuda", 0 if one_dev else local_rank)
    torch.cuda.set_device(dev)
    red_dev = torch.device("cpu") if one_dev else dev   # where the scalars of the control collectives live
    affinity = pin_to_gpu_numa(dev)
    log(f"[rank {rank}] host affinity: {affinity}")
    rccl = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)   # "nccl" is RCCL on ROCm
        names = [None] * world
        dist.all_gather_object(names, f"rank{rank}:cuda{local_rank}:{torch.cuda.get_device_name(dev)}")
        rccl = {"backend": dist.get_backend(), "ranks_seen": dist.get_world_size(), "devices": names}

    from sgs_hip import raster, _lib
    from sgs_hip.synthetic import CONFIGS, make_scene

    P0, C0, W, H, fx = CONFIGS[args.config]
    P = args.points or P0
    C = args.channels or C0
    t0 = time.time()
    scene = make_scene(P, C, W, H, fx, seed=0)
    V = max(1, args.views)
    # slot i of rank r cycles through NCAM cameras: camera (slot, k) is view number (r * V + i) * NCAM + k
    cams_host = [[view_camera((rank * V + i) * NCAM + k, W, H, fx) for k in range(NCAM)] for i in range(V)]
    cam = cams_host[0][0]
    log(f"[rank {rank}] scene P={P} C={C} {W}x{H} generated in {time.time() - t0:.1f}s")
    s = scene.to(dev)
    cams = [[cm.to(dev) for cm in row] for row in cams_host]
    empty = torch.Tensor([])

    # inference: state buffers stay resident (as under torch.no_grad); one pool and stream per view in flight
    pools = [raster.ScratchPool() for _ in range(V)]
    plain_streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(V - 1)]
    streams = plain_streams
    # the headline's streams: per view slot a blend stream on CUs [front_cus, n) with its front end on CUs [0, front_cus)
    part = raster.PartitionedStreams(dev, args.front_cus, V, blend_everywhere=args.blend_everywhere) if args.front_cus > 0 else None
    if part is None and args.stream_priority != "none":
        class _PriorityStreams:
            def __init__(self, high):
                import ctypes
                from sgs_hip import _lib
                pb, pf = {"split": (0, 0), "blend": (-1, 0), "front": (0, -1)}[high]
                self.streams = [torch.cuda.Stream(dev, priority=pb) for _ in range(V)]
                self.front = [torch.cuda.Stream(dev, priority=pf) for _ in range(V)]
                for b_, f_ in zip(self.streams, self.front):
                    _lib.check(_lib.load().sgs_stream_set_front(ctypes.c_void_p(b_.cuda_stream), ctypes.c_void_p(f_.cuda_stream)), "set front")
        part = _PriorityStreams(args.stream_priority)

    def render(i, deferred=False, k=0):
        c = cams[i][k % NCAM]
        fn = raster.rasterize_forward_deferred if deferred else raster.rasterize_forward
        with torch.cuda.stream(streams[i]):
            return fn(s.bg, s.means3D, s.features, s.opacities, s.scales, s.rotations, 1.0, empty,
                      c.world_view_transform, c.full_proj_transform, c.tanfovx, c.tanfovy, H, W, empty, 0,
                      c.camera_center, False, False, C, False, pool=pools[i])

    # host pattern of a forward: "speculative" (default: raster.rasterize_forward_inference), "classic" (mid-frame wait), True = deferred counts
    DEFER = True if args.deferred_count else ("classic" if args.classic_count else "speculative")

    def step(k=0, defer=None, prev=None):   # one batch: V views of the scene (camera k of every slot), all in flight together
        # `prev`: the previous step's results.  Slot i's old feature map is released right before slot i renders again: the
        # block goes back to that stream's pool and the new frame reuses it in stream order, so a step keeps V feature maps
        # alive, not 2 V (round 3: 45 GB reserved for four views in flight -- half of it was the previous step's outputs).
        def drop(i):
            if prev is not None:
                prev[i] = None
        mode = DEFER if defer is None else defer
        if mode in (False, "classic"):   # the reference's host pattern: every forward blocks on its num_rendered read-back in the middle of the frame
            res = []
            for i in range(V):
                drop(i)
                res.append(render(i, False, k))
            return res
        if mode == "speculative":   # every forward still returns its num_rendered (the host waits for it), but only after the whole frame is enqueued
            res = []
            for i in range(V):
                drop(i)
                res.append(render(i, True, k).result())
            return res
        # deferred counts (SGS_OPT_DEFER_COUNT): all V forwards are enqueued without the host waiting for the GPU,
        # then every frame's counts are checked (a frame that outgrew its capacity guess is rendered again there)
        pending = []
        for i in range(V):
            drop(i)
            pending.append(render(i, True, k))
        return [h.result() for h in pending]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def single_view(variant, n=24, deferred=False, inference=False, stage_events=True):
        """One view in flight: per-forward device time (hipEvents on the launch stream = torch's current
        stream) and the per-stage times (deferred resolution: no extra synchronisation).  deferred: the forward
        without the num_rendered read-back (each frame's counts are checked before the next one is enqueued)."""
        raster.set_blend_variant(variant)
        # warm up for 0.4 s, not for 3 frames: the same kernel runs 5-7 % slower in the first ~0.2 s of GPU activity of a
        # process than in steady state (clock ramp; tools/exp_r03_sweep2.py with a repeated variant shows the drift)
        t_w = time.perf_counter()
        nw = 0
        while nw < 3 or time.perf_counter() - t_w < 0.4:
            render(0)
            nw += 1
            if nw % 8 == 0:
                torch.cuda.synchronize(dev)
        torch.cuda.synchronize(dev)
        raster.get_stage_ms()
        raster.set_stage_timing(2 if stage_events else 0)   # (the eight per-stage event records of a frame are ~45 us of launch gaps)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, b in evs:
            a.record()
            if inference:   # the whole frame enqueued against the capacity guess, THEN the host waits for the counts (raster.rasterize_forward_inference)
                h = render(0, True).result()
            else:
                h = render(0, deferred)
            b.record()
            if deferred:
                h.result()
        torch.cuda.synchronize(dev)
        raster.set_stage_timing(0)
        ms = sorted(a.elapsed_time(b) for a, b in evs)
        stage = raster.get_stage_ms()
        med = ms[len(ms) // 2]
        return dict(value=H * W * C / (med * 1e-3) / 1e9, unit="Gpixel*channels/s", ms_median=med,
                    ms_mean=sum(ms) / len(ms), ms_min=ms[0], forwards=n,
                    stage_ms=dict(zip(STAGES, [round(v, 4) for v in stage]))), stage

    def deferred_retries():
        n = 0
        for st in streams:
            with torch.cuda.stream(st):
                n += raster.stream_stat(_lib.STAT_DEFERRED_RETRIES, dev)
        return n

    rank_ms = {}   # ms per step of the slowest / fastest rank in the last in_flight() (a straggler shows here)

    def in_flight(variant, steps, warmup, stage_timing=False, defer=None, partitioned=True):
        nonlocal streams
        raster.set_blend_variant(variant)
        streams = part.streams if (part is not None and partitioned) else plain_streams
        for st_ in streams:
            st_.wait_stream(torch.cuda.current_stream(dev))
        try:
            return _in_flight(variant, steps, warmup, stage_timing, defer)
        finally:
            torch.cuda.synchronize(dev)
            streams = plain_streams

    def _in_flight(variant, steps, warmup, stage_timing, defer):
        out = None
        for k in range(warmup):
            out = step(k, defer, out)   # exactly like the timed loop, so that the caching allocator reaches its steady-state
            #                footprint here, not in the timed region (a 10 GB hipMalloc in timed step 2 was a 240 ms stall,
            #                profiles/r02h_bench_default.json)
        if stage_timing:   # deferred per-stage hipEvents for the timed steps only (no extra synchronisation)
            raster.get_stage_ms()
            raster.set_stage_timing(2)
        retries0 = deferred_retries()
        gc.collect()
        gc.disable()   # a generation-2 collection of the interpreter's heap is a 50 ms host pause (measured)
        barrier()
        t0 = time.perf_counter()
        marks, mism = [], 0
        for k in range(steps):
            out = step(k, defer, out)
            mism += sum(int(o[0] != ref_n[i][k % NCAM]) for i, o in enumerate(out))   # host ints, no device work
            marks.append(time.perf_counter())
        barrier()
        t = time.perf_counter() - t0
        gc.enable()
        retries = deferred_retries() - retries0
        rank_ms["min"] = rank_ms["max"] = t / steps * 1e3   # this rank's own clock; over the ranks below
        if world > 1:
            tt = torch.tensor([t, -t], device=red_dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            rank_ms["max"], rank_ms["min"] = float(tt[0].item()) / steps * 1e3, -float(tt[1].item()) / steps * 1e3
            t = float(tt[0].item())
        per = [b - a for a, b in zip([t0] + marks[:-1], marks)]
        per.sort()
        if rank == 0:   # host-side enqueue cadence (diagnostic only; the metric uses t / steps)
            log(f"variant {variant}: per-step host ms min/median/max: {per[0] * 1e3:.2f} / {per[len(per) // 2] * 1e3:.2f} / {per[-1] * 1e3:.2f}")
        return t / steps * 1e3, per[len(per) // 2] * 1e3, mism, out, retries

    # ---- one view in flight (SURVEY 8(d)'s t_fwd), default arithmetic; its stage times feed the roofline
    sv_cold, stage_ms_cold = single_view(args.variant)   # (in the first second of the process's GPU activity: see the steady-state leg behind the headline)
    # the same through the inference entry point (what the drop-in module does under no_grad): no read-back hole in the GPU's timeline
    sv_inference = single_view(args.variant, inference=True)[0]
    # ... and as the product runs it: without the library's per-stage event records (instrumentation the two legs above switch on)
    sv_untimed = single_view(args.variant, inference=True, stage_events=False)[0]
    sv_inference["ms_median_without_stage_events"] = sv_untimed["ms_median"]
    sv_inference["value_without_stage_events"] = sv_untimed["value"]
    # integrity reference: every (slot, camera)'s num_rendered, rendered alone (concurrent forwards must reproduce it)
    ref_n = []
    for i in range(V):
        row = []
        for k in range(NCAM):
            row.append(render(i, False, k)[0])
            torch.cuda.synchronize(dev)
        ref_n.append(row)
    if world > 1:   # a mis-sharded run shows at a glance: every rank's device, its view numbers and their num_rendered
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"rank": rank, "device": f"cuda{local_rank}:{torch.cuda.get_device_name(dev)}", "host_affinity": affinity,
                                          "views": [(rank * V + i) * NCAM for i in range(V)],
                                          "num_rendered_first_camera": [row[0] for row in ref_n]})
        rccl["per_rank"] = per_rank

    # ---- what a user of the reference gets: the drop-in module, called as model/renderer.py:169-185,228 calls it
    def api_path(n=24, debug=True):
        import channel_rasterization as chn
        c = cams[0][0]
        prm = [torch.nn.Parameter(t) for t in (s.means3D, s.features, s.opacities, s.scales, s.rotations)]
        settings = chn.GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=s.bg, scale_modifier=1.0,
            viewmatrix=c.world_view_transform, projmatrix=c.full_proj_transform, sh_degree=0, campos=c.camera_center,
            prefiltered=False, debug=debug, num_channels=C)
        ras = chn.GaussianRasterizer(raster_settings=settings)
        raster.set_blend_variant(args.variant)

        def call():
            screenspace = torch.zeros_like(prm[0], requires_grad=False)
            return ras(means3D=prm[0], means2D=screenspace, shs=None, colors_precomp=prm[1], opacities=prm[2],
                       scales=prm[3], rotations=prm[4], cov3D_precomp=None)
        with torch.no_grad():
            for _ in range(3):
                call()
            torch.cuda.synchronize(dev)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
            for a, b in evs:
                a.record()
                out_ = call()
                b.record()
            torch.cuda.synchronize(dev)
        ms = sorted(a.elapsed_time(b) for a, b in evs)
        med = ms[len(ms) // 2]
        return dict(value=H * W * C / (med * 1e-3) / 1e9, unit="Gpixel*channels/s", ms_median=med, ms_min=ms[0], forwards=n,
                    call="channel_rasterization.GaussianRasterizer(settings)(means3D, means2D, opacities, colors_precomp, scales, "
                         "rotations) under torch.no_grad(), nn.Parameter inputs, debug=True (model/renderer.py:169-185,228), "
                         "one view, output + radii allocated per call; the host waits for num_rendered before the call returns, after the "
                         "whole frame was enqueued against the stream's capacity guess (sgs_hip.api.SPECULATIVE_COUNT, inference only)",
                    out_shape=list(out_[0].shape))
    # ---- the headline: K timed steps, V views in flight, default arithmetic
    gc.collect()
    torch.cuda.synchronize(dev)
    for p_ in pools:
        p_.clear()
    torch.cuda.empty_cache()              # (the legs above leave their own cached blocks behind: the memory figures below are
    torch.cuda.reset_peak_memory_stats(dev)   # the headline's -- scene, resident state buffers, V feature maps)
    ms_per_step, ms_step_median, mismatches, out, retries = in_flight(args.variant, args.steps, max(2, args.warmup), True)
    raster.set_stage_timing(0)
    stage_ms_timed = raster.get_stage_ms()   # per-stream event spacing inside the timed region (queueing included)
    headline_rank_ms = dict(rank_ms)
    mem = {"max_allocated_GB": round(torch.cuda.max_memory_allocated(dev) / 1e9, 2),
           "reserved_GB": round(torch.cuda.memory_reserved(dev) / 1e9, 2),
           "scene_GB": round(sum(t.numel() * t.element_size() for t in (s.means3D, s.features, s.opacities, s.scales, s.rotations)) / 1e9, 2),
           "note": f"this rank, up to the end of the headline's timed region ({V} views in flight: per view slot one feature map "
                   f"of {C * H * W * 4 / 1e9:.2f} GB + resident geometry / binning / work-list buffers; a slot's previous map is "
                   "released before it renders again); the legs measured afterwards add their own buffers"}
    mem["per_view_slot_GB"] = round((mem["max_allocated_GB"] - mem["scene_GB"]) / V, 2)
    if rank == 0:
        log(f"torch max allocated {mem['max_allocated_GB']:.2f} GB, reserved {mem['reserved_GB']:.2f} GB")

    # ---- one view in flight in STEADY STATE: the same leg as sv_cold, measured right behind the timed region (seconds of continuous GPU work).
    # Round 6: 24 forwards at the top of the process are a 40-ms window, and the sweep's duration wanders by +-3 % over a process's life on these
    # boxes (call G: 1.003 ms there against 0.946 in the legs behind the headline; call I: 0.97 there against 1.00 right behind it -- clocks under
    # a changing load).  The roofline is therefore quoted on 200 forwards right behind the timed region, the state the timed region runs in; the short
    # leg at the top stays in the line as `single_view_cold`, and the rocprofv3 average over a whole process (profiles/) is the cross-check.
    for p_ in pools:
        p_.clear()
    sv_default, stage_ms = single_view(args.variant, n=200)   # (0.3 s of single-view work: a window long enough to average over the clock's excursions)
    api = api_path() if world == 1 or rank == 0 else None
    if api is not None:
        api["ratio_to_single_view"] = api["ms_median"] / sv_default["ms_median"]   # (both right behind the timed region)
        api["ratio_to_single_view_inference"] = api["ms_median"] / sv_inference["ms_median"]
        api["ms_median_with_debug_false"] = api_path(debug=False)["ms_median"]
        api["note"] = ("the events bracket the Python call: the span includes the host's own work before the first launch and "
                       "after the last (module call, autograd Function, ctypes marshalling, with debug=True one end-of-call "
                       "synchronisation) during which the GPU idles; single_view brackets the internal entry point")

    # ---- workload statistics of this rank's first view (slot 0, camera 0: the view single_view / the roofline time;
    # every run prints them: bytes depend on them)
    del out
    raster.set_blend_variant(args.variant)
    out = [render(0, False, 0)]
    torch.cuda.synchronize(dev)
    num_rendered, color, radii, geom, binn, img, _ = out[0]
    iv = raster.image_views(img, W, H)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    nc = torch.zeros(gy * 16, gx * 16, dtype=torch.int32, device=dev)
    nc[:H, :W] = iv["n_contrib"]
    n_eff = nc.view(gy, 16, gx, 16).permute(0, 2, 1, 3).reshape(gy * gx, 256).amax(dim=1)
    sum_neff = int(n_eff.sum().item())
    contributors = int(iv["n_contrib"].sum().item())   # upper bound on per-pixel contributors
    ranges = iv["ranges"]
    lens = (ranges[:, 1] - ranges[:, 0]).to(torch.float64)
    p_vis = int((radii > 0).sum().item())
    tiles = gx * gy
    n_eff_host = n_eff.cpu().numpy().astype(np.int64)
    bytes_blend = 4 * C * H * W + (4 * C + 28) * sum_neff + 8 * H * W + 8 * tiles
    bytes_front = 44 * P + 36 * p_vis + 36 * num_rendered
    flops_alg = 2.0 * C * contributors + 20.0 * 256 * sum_neff   # accumulate + per-entry weight chain
    blend_ms = stage_ms[5] + stage_ms[6]   # weights pre-pass + accumulate (one kernel each on the default path)
    achieved = bytes_blend / (blend_ms * 1e-3) if blend_ms > 0 else 0.0
    del out, color

    # ---- other arithmetics / host patterns, the backward (N = 1 extras; skipped under --no-extras)
    exact = backward = two_term = deferred = classic = None
    kx = max(8, min(args.steps, 60))

    def extra_leg(variant, what, defer=None, sv=True, partitioned=True):
        sv_, stg = single_view(variant, deferred=defer is True) if sv else (None, None)
        ms_e, ms_e_med, mism_e, out_e, retr = in_flight(variant, kx, max(3, NCAM + 2), defer=defer, partitioned=partitioned)   # every camera of a slot and the wrap-around seen once: buffers at their steady size
        del out_e
        return {"arithmetic": what, "value": world * V * H * W * C / (ms_e * 1e-3) / 1e9, "unit": "Gpixel*channels/s",
                "ms_per_step": ms_e, "ms_per_view": ms_e / V, "steps": kx, "views_in_flight": V,
                "cu_partition": bool(part is not None and partitioned),
                "num_rendered_mismatches_vs_serial": mism_e, "deferred_retries": retr, "single_view": sv_,
                "roofline_frac": (bytes_blend / ((stg[5] + stg[6]) * 1e-3) / HBM_PEAK) if sv and stg[5] + stg[6] > 0 else None}
    # the reference's own host pattern (the wait for num_rendered in the MIDDLE of the frame) always rides beside the headline, and so
    # does round 5's stream arrangement (ordinary streams, every kernel on any compute unit)
    classic = extra_leg(args.variant, "default arithmetic, the host waits for num_rendered in the MIDDLE of every forward (the reference's host "
                                      "pattern, rasterizer_impl.cu:283; rounds 1-4's headline)", defer="classic", sv=not args.no_extras)
    shared_cus = None
    if part is not None:
        shared_cus = extra_leg(args.variant, "default arithmetic and host pattern on ORDINARY streams: every kernel of every view on any compute unit "
                                             "(round 5's headline arrangement)", sv=False, partitioned=False)
    if not args.no_extras:
        exact = extra_leg(EXACT, "fp32-input MFMA (v_mfma_f32_32x32x2_f32), fp32 accumulate: bit-identical to the contract")
        two_term = extra_leg(TWO_TERM, "round 2's default: two bf16 terms per operand, three MFMA products "
                                       "(<= 3 * 2^-16 of sum |f| w per term): NOT fp32-class, kept selectable (blend variant 14)")
        deferred = extra_leg(args.variant, "default arithmetic, deferred counts (SGS_OPT_DEFER_COUNT, inference only): no host "
                                           "read-back inside the forward", defer=True)
        raster.set_blend_variant(args.variant)
    consumer = None
    if not args.no_extras and world == 1 and C % 128 == 0:
        # what the reference does with the map (eval_segmentation.py:155-157), three ways
        from sgs_hip import semantic
        import channel_rasterization as chn
        c = cams[0][0]
        n_cls = 21   # ScanNet-20 prompts + "other"
        text = torch.nn.functional.normalize(torch.randn(n_cls, C, device=dev, generator=torch.Generator(dev).manual_seed(7)), dim=1)
        settings = chn.GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=s.bg, scale_modifier=1.0,
            viewmatrix=c.world_view_transform, projmatrix=c.full_proj_transform, sh_degree=0, campos=c.camera_center,
            prefiltered=False, debug=False, num_channels=C)
        proj = semantic.project_features(s.features, text)
        geo = (s.means3D, s.opacities, s.scales, s.rotations)

        def ref_flow():
            r = chn.GaussianRasterizer(settings)(means3D=s.means3D, means2D=torch.zeros_like(s.means3D), opacities=s.opacities,
                                                 colors_precomp=s.features, scales=s.scales, rotations=s.rotations)[0]
            r = r / (r.norm(dim=0, keepdim=True) + 1e-8)
            return torch.einsum("cq,qhw->chw", text, r)

        def timed(fn, n=12):
            with torch.no_grad():
                for _ in range(2):
                    o = fn()
                torch.cuda.synchronize(dev)
                evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
                for a, b in evs:
                    a.record()
                    o = fn()
                    b.record()
                torch.cuda.synchronize(dev)
            ms = sorted(a.elapsed_time(b) for a, b in evs)
            return ms[len(ms) // 2], o
        t_ref, sim_ref = timed(ref_flow)
        t_norm, sim_n = timed(lambda: semantic.render_similarity(settings, *geo, s.features, text, normalised=True, projected=proj))
        t_log, _ = timed(lambda: semantic.render_similarity(settings, *geo, s.features, text, projected=proj))
        consumer = {"what": f"per-view similarities with {n_cls} text embeddings (eval_segmentation.py:155-157), cfg3 scene",
                    "reference_flow_ms": t_ref, "normalised_via_norm_plane_ms": t_norm, "unnormalised_logits_ms": t_log,
                    "max_abs_diff_normalised_vs_reference_flow": float((sim_n - sim_ref).abs().max().item()),
                    "note": "reference flow = render the (C,H,W) map with this library, then the reference's two torch lines; "
                            "norm-plane flow = the n_cls-channel projected render + one more blend whose epilogue adds sum_c out^2 "
                            "into an (H,W) plane instead of storing the map (SGS_OPT_NORM_PLANE)"}
        del sim_ref, sim_n, proj, text
        torch.cuda.empty_cache()
    if not args.no_extras and world == 1:
        for p in pools:
            p.clear()
        torch.cuda.empty_cache()
        c = cams[0][0]
        dL = torch.randn(C, H, W, device=dev)
        ev = lambda: torch.cuda.Event(enable_timing=True)   # noqa: E731

        def fwd_bwd(marks=None):
            n, col, rad, g_, b_, i_, _ = raster.rasterize_forward(
                s.bg, s.means3D, s.features, s.opacities, s.scales, s.rotations, 1.0, empty,
                c.world_view_transform, c.full_proj_transform, c.tanfovx, c.tanfovy, H, W, empty, 0,
                c.camera_center, False, False, C, False)
            if marks is not None:
                marks[1].record()
            raster.rasterize_backward(s.bg, s.means3D, rad, s.features, s.scales, s.rotations, 1.0, empty,
                                      c.world_view_transform, c.full_proj_transform, c.tanfovx, c.tanfovy, dL,
                                      empty, 0, c.camera_center, g_, n, b_, i_, False)
        for _ in range(2):
            fwd_bwd()
        torch.cuda.synchronize(dev)
        evs = [(ev(), ev(), ev()) for _ in range(8)]
        for m in evs:
            m[0].record()
            fwd_bwd(m)
            m[2].record()
        torch.cuda.synchronize(dev)
        ms = sorted(m[0].elapsed_time(m[2]) for m in evs)
        ms_b = sorted(m[1].elapsed_time(m[2]) for m in evs)
        # algorithmic bytes of the backward: the gradient read once, the active feature rows read and their gradient
        # rows written once each, the list data, final_T / n_contrib, the per-Gaussian geometry gradients
        bytes_bwd = 4 * C * H * W + 2 * (4 * C) * sum_neff + 28 * sum_neff + 8 * H * W + 8 * tiles + (44 + 60) * p_vis
        t_b = ms_b[len(ms_b) // 2]
        backward = {"workload": "cfg3 forward + backward (dL/dout random), C = 512: the reference's backward is "
                                "compiled for 3 channels only",
                    "fwd_bwd_ms_median": ms[len(ms) // 2], "fwd_bwd_ms_min": ms[0],
                    "backward_ms_median": t_b,
                    "roofline": {"bound": "hbm", "algorithmic_bytes": bytes_bwd, "achieved": bytes_bwd / (t_b * 1e-3) / 1e9,
                                 "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": bytes_bwd / (t_b * 1e-3) / HBM_PEAK,
                                 "note": "4 C H W (dL/dpixel once) + 2 x 4 C sum n_t_eff (feature rows in, colour-gradient rows "
                                         "out) + 28 sum n_t_eff + 8 H W + 104 P_vis; since round 5 one kernel reads the gradient once per "
                                         "128 entries of a tile for BOTH products (DESIGN.md 5.5, 5.14); fabric bytes by PMC: "
                                         "profiles/r06_backward_pmc.txt"},
                    "includes": "output / gradient allocation through the caching allocator (no resident pool: the "
                                "state buffers belong to the autograd graph)"}
        del dL
        torch.cuda.empty_cache()

    for p in pools:   # (the headline's residents are no longer needed)
        p.clear()
    raster.INFERENCE_POOL.clear()
    torch.cuda.empty_cache()
    if rank == 0:
        log(f"P_vis={p_vis} L={num_rendered} tile-list mean/max={lens.mean().item():.1f}/"
            f"{int(lens.max().item())} sum_n_t_eff={sum_neff} (mean {sum_neff / tiles:.1f}/tile) "
            f"sum_n_contrib={contributors}")
        log("stage ms (one view in flight): " + ", ".join(f"{n}={v:.3f}" for n, v in zip(STAGES, stage_ms)))
        log(f"bytes_alg: blend {bytes_blend / 1e9:.3f} GB + front-end {bytes_front / 1e9:.3f} GB; "
            f"whole-forward HBM fraction {V * (bytes_blend + bytes_front) / (ms_per_step * 1e-3) / HBM_PEAK:.3f}")
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "blend_traffic.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                if tj.get("config") == args.config and tj.get("variant", 0) == args.variant:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:   # noqa: BLE001
                traffic = None
        split = C >= 128 and args.variant not in (EXACT,)
        arith = {0: "f32-equivalent (3-term bf16 splits, 6 MFMA products, f32 accumulate)", TWO_TERM: "bf16x3-split products, f32 accumulate"}.get(args.variant, f"blend variant {args.variant:#x}")
        res = {
            "metric": "Gpixel*channels/s forward render (1M Gauss, C=512, 968x1296)",
            "value": world * V * H * W * C / (ms_per_step * 1e-3) / 1e9,
            "unit": "Gpixel*channels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": (arith if split else "f32"),
            "data": "synthetic",
            "config": {"workload": f"{args.config}: P={P} Gaussians, C={C}, {H}x{W} forward render "
                                   f"(BASELINE.md config 3 generator, seed 0)",
                       "views_per_step_per_gpu": V, "hip_streams_per_gpu": V * (2 if part is not None else 1),
                       "parallelism": f"views x{world * V}: {V} in flight per GPU on {V} HIP streams, {world} GPU(s), "
                                      f"scene replicated, no collective",
                       "rccl": rccl,
                       "cameras": f"each of the {V} view slots cycles through {NCAM} cameras: a different view every step",
                       "num_rendered": ("blocking read-back in the middle of every forward (the reference's host pattern, rasterizer_impl.cu:283)" if DEFER == "classic" else
                                        "every forward returns its num_rendered as the reference's does (the host waits for it), after the whole frame was "
                                        "enqueued against the stream's capacity guess: raster.rasterize_forward_inference, what the drop-in module does under "
                                        "torch.no_grad() (a frame that outgrew the guess is rendered again; every num_rendered is compared with the serial render); "
                                        "classic_count below = rounds 1-4's headline pattern" if DEFER == "speculative" else
                                        "deferred (SGS_OPT_DEFER_COUNT): buffers sized from the stream's previous frame, counts "
                                        "checked on the device and on the host once per step; every step's num_rendered is "
                                        "compared with the serial render (integrity)"),
                       "blend_variant": args.variant,
                       "blend_arithmetic": ("fp32 MFMA, bit-exact" if not split else
                                            "every weight, decision and integer output in contract fp32 (bit-exact); the C >= 128 "
                                            "weighted sum with features and weights split EXACTLY into three bf16 terms each, the six "
                                            "products with i + j <= 4 on v_mfma_f32_32x32x16_bf16 (round 5: the double-rate MFMA in the ping-pong sweep, whose workgroups own "
                                            "their compute units -- DESIGN.md 5.10), fp32 accumulate.  tests: against the "
                                            "exact (float64) composite it is as accurate as the oracle's fp32 fma chain (max and rms "
                                            "of the element-wise error |x - exact| / max(|exact|, 1e-3 |pixel|_inf)); against the oracle "
                                            "itself |out - oracle| <= 5e-6 |pixel|_inf everywhere, and the element-wise form exceeds 1e-4 "
                                            "on < 1e-5 of the elements -- as the oracle's own chain does against the exact composite "
                                            "(tests/test_configs_gpu.py); exact_f32 below is the bit-identical fp32-MFMA path")},
            "ms_per_step_median": ms_step_median,
            "host_affinity": affinity,
            "ms_per_step_ranks": headline_rank_ms,   # slowest / fastest rank's own clock over the same timed region
            "memory": mem,
            "ms_per_view": ms_per_step / V,
            "single_view": sv_default,
            "single_view_cold": dict(sv_cold, note="the same leg at the top of the process, 0.4 s into its GPU activity (what rounds 1-5 reported as single_view)"),
            "single_view_inference": dict(sv_inference, note="one view in flight through raster.rasterize_forward_inference: the whole frame is "
                                          "enqueued against the stream's capacity guess before the host waits for num_rendered (what "
                                          "GaussianRasterizer does under torch.no_grad(); single_view above keeps the reference's mid-frame wait)"),
            "api_path": api,
            "host_pattern": {"speculative": "speculative: the frame is enqueued in full against the stream's capacity guess, then the host waits for its num_rendered",
                             "classic": "classic: the host waits for num_rendered in the middle of every forward (the reference's pattern)",
                             True: "deferred counts"}[DEFER],
            "cu_partition": ({"front_cus": part.front_cus, "blend_cus": part.cu_count - part.front_cus, "device_cus": part.cu_count,
                              "note": "every view slot: blend stream confined to the blend CUs, front end (preprocess, depth sort, span partitions) "
                                      "on a second stream confined to the front CUs (hipExtStreamCreateWithCUMask; sgs_stream_set_front); "
                                      "shared_cus below = the same headline on ordinary streams"} if (part is not None and hasattr(part, "front_cus")) else
                             ({"stream_priority": args.stream_priority, "note": "experiment: front ends on second ordinary streams (--stream-priority)"} if part is not None else None)),
            "shared_cus": shared_cus,
            "classic_count": classic,
            "deferred_count": deferred,
            "exact_f32": exact,
            "two_term": two_term,
            "backward": backward,
            "semantic_consumer": consumer,
            # the forward blend = blend_weights2_sb_kernel + blend_accum_sweep3_kernel (one launch each);
            # SURVEY 8(d)'s algorithmic bytes are a property of the pair, so the roofline is quoted
            # on the pair; the per-kernel live durations are alongside (rocprof: profiles/).
            "roofline": {"bound": "hbm", "kernel": "blend_fwd (blend_weights2_sb_kernel + blend_accum_sweep3_kernel)",
                         "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": traffic, "algorithmic_bytes": bytes_blend,
                         "kernel_ms": blend_ms,
                         "kernels_ms": {"blend_weights": round(stage_ms[5], 4), "blend_accum": round(stage_ms[6], 4)},
                         "secondary_ceilings": {
                             "algorithmic_gflop": flops_alg / 1e9,
                             "fp32_fma_frac": (flops_alg / (blend_ms * 1e-3) / FP32_FMA_PEAK) if blend_ms > 0 else None,
                             "bf16_mfma_frac_6_products": (6 * 2.0 * C * 256 * sum_neff * 0.55 / (blend_ms * 1e-3) / BF16_MFMA_PEAK) if blend_ms > 0 else None,
                             "note": "algorithmic flops / blend time against the fp32 vector (= fp32 MFMA) peak 157.3 TF; the six "
                                     "bf16 products (all 256 pixels of every active entry, ~55 % of sum n_t_eff) against the "
                                     "dense bf16 peak: since round 5 the sweep issues the double-rate 32x32x16 instruction "
                                     "(1024 FLOP/clk/SIMD; DESIGN.md 5.10 for why that needed eight-wave workgroups that own "
                                     "their compute unit) -- PMC: profiles/r06_blend_pmc.txt (SQ_VALU_MFMA_BUSY_CYCLES)"},
                         "measured": f"hipEvents on the launch stream over {sv_default['forwards']} forwards with one view "
                                     f"in flight, right behind the timed region (steady state; the same leg at the top of the process: "
                                     f"single_view_cold); the timed region keeps {V} in flight (stage_ms_timed_region)"},
            # bytes_alg of the WHOLE forward (blend + binning front end) over the frame time
            "whole_forward": {"algorithmic_bytes": bytes_blend + bytes_front,
                              "achieved_GBps": V * (bytes_blend + bytes_front) / (ms_per_step * 1e-3) / 1e9,
                              "frac_of_hbm_peak": V * (bytes_blend + bytes_front) / (ms_per_step * 1e-3) / HBM_PEAK,
                              "single_view_frac_of_hbm_peak": (bytes_blend + bytes_front) / (sv_default["ms_median"] * 1e-3) / HBM_PEAK},
            "stage_ms": dict(zip(STAGES, [round(v, 4) for v in stage_ms])),
            "stage_ms_timed_region": dict(zip(STAGES, [round(v, 4) for v in stage_ms_timed])),
            "multi_gpu_configs": None,
            "integrity": {"forwards_checked": args.steps * V, "num_rendered_mismatches_vs_serial": mismatches,
                          "deferred_retries": retries},
            "workload_stats": {"P_vis": p_vis, "num_rendered": num_rendered, "sum_n_t_eff": sum_neff,
                               "tiles": tiles, "sum_n_contrib": contributors},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline_pytorch(scene, cam, C, W, H, n_eff_host)
            log("cpu_baseline: " + res["cpu_baseline"]["sample"])
            res["cpu_baseline_port"] = cpu_baseline_port(scene, cam, C, W, H, n_eff_host)
            log("cpu_baseline_port: " + res["cpu_baseline_port"]["sample"])
    if args.extra_configs or world > 1:
        # BASELINE configs 4 / 5 on this job's ranks, AFTER the line above is complete and under a watchdog: the RCCL band
        # exchange of config 5 has never run on more than one GPU (this pool hands out single-GPU boxes), and a collective
        # that hangs must not cost the job its headline -- after EXTRA_TIMEOUT_S rank 0 prints the line without them
        import threading
        finished = threading.Event()

        def watchdog():
            if finished.wait(EXTRA_TIMEOUT_S):
                return
            if rank == 0:
                res["multi_gpu_configs"] = {"error": f"configs 4 / 5 did not finish within {EXTRA_TIMEOUT_S} s; the rest of the line is complete"}
                print(json.dumps(res), flush=True)
            os._exit(0)
        threading.Thread(target=watchdog, daemon=True).start()
        extra_cfgs = extra_config_legs(rank, world, dev, dist if world > 1 else None, red_dev)
        finished.set()
        if rank == 0:
            log("extra configs: " + json.dumps(extra_cfgs)[:600])
            res["multi_gpu_configs"] = extra_cfgs
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

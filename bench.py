#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X N-channel Gaussian-splat rasteriser.

Metric (BASELINE.json): Gpixel*channels/s of the FORWARD feature render,
    1M synthetic Gaussians, C = 512, 968x1296   ("cfg3")
  = H*W*C / t_fwd / 1e9, t_fwd = the whole rasterize_gaussians forward
    (preprocess -> scan -> key emission -> 64-bit radix sort -> tile ranges -> blend),
    inputs already resident in HBM, called through the C-ABI.

A "step" is one forward render of one view.  With --gpus N (one process per GPU, launched
by torch.distributed.run) the scene is replicated and every rank renders its own view per
step -- views shard embarrassingly, there is no data-path collective -- so per-GPU work is
fixed ("scaling": "weak") and `value` is the whole-job aggregate.

The single JSON line also carries
  roofline     : the dominant kernel (blend forward) against the HBM roofline.  achieved =
                 algorithmic bytes per launch (4CHW + (4C+28)*sum_t n_t_eff +
                 8HW + 8*tiles) / mean kernel duration measured with hipEvents on the launch
                 stream INSIDE the timed region (deferred resolution, no extra sync);
  cpu_baseline : the CPU oracle (a C port of the algorithm, OpenMP over tiles) timed on this
                 host on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
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

HBM_PEAK = 8.0e12   # MI355X spec, MI355X_MICROARCH.md "Chip-level parameters"


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def view_camera(rank, W, H, fx):
    """Rank r looks at the same slab from a slightly shifted / yawed position so that every
    rank has a full-sized but different view."""
    import math
    from sgs_hip.camera import make_camera, focal2fov
    a = 0.04 * rank
    R = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    T = np.array([0.02 * rank, -0.01 * rank, 0.0])
    return make_camera(R, T, focal2fov(fx, W), focal2fov(fx, H), W, H)


def cpu_baseline(scene, cam, C, W, H, budget_s=20.0):
    """Oracle timed on the host cores: preprocess + binning in full, blend on a bounded tile
    sample extrapolated by the tiles' list work (sum n_t_eff of the sample vs the frame)."""
    import ctypes as Ct
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
    # sample whole tile rows from the middle of the frame until the budget is used
    lo = (gy // 2) * gx
    n_sample = min(ntiles - lo, max(gx, 2 * nthreads))
    t_blend, done = 0.0, 0
    while True:
        t0 = time.time()
        orc.blend_forward(pre, binn, feats, scene.bg.numpy(), W, H, tile_lo=lo + done,
                          tile_hi=lo + done + n_sample)
        t_blend += time.time() - t0
        done += n_sample
        if t_blend > budget_s or lo + done + n_sample > ntiles:
            break
    frac = done / ntiles
    t_frame = t_front + t_blend / frac
    return dict(value=H * W * C / t_frame / 1e9, unit="Gpixel*channels/s", cores=nthreads,
                kind="port",
                sample=(f"oracle (C port, OpenMP x{nthreads}): preprocess+binning in full "
                        f"({t_front:.2f} s) + blend on {done} of {ntiles} tiles "
                        f"({t_blend:.2f} s) extrapolated by tile count -> {t_frame:.1f} s/frame"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--points", type=int, default=None, help="override P (debug)")
    ap.add_argument("--channels", type=int, default=None, help="override C (debug)")
    ap.add_argument("--variant", type=int, default=0, help="blend kernel variant (tuning)")
    ap.add_argument("--views", type=int, default=4,
                    help="views in flight per GPU: a step renders this many views of the scene, one per HIP "
                         "stream, so one view's front-end and host round trip overlap another view's blend")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks "
                         f"(WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)   # "nccl" is RCCL on ROCm

    from sgs_hip import raster
    from sgs_hip.synthetic import CONFIGS, make_scene

    P0, C0, W, H, fx = CONFIGS[args.config]
    P = args.points or P0
    C = args.channels or C0
    t0 = time.time()
    scene = make_scene(P, C, W, H, fx, seed=0)
    V = max(1, args.views)
    cams_host = [view_camera(rank * V + i, W, H, fx) for i in range(V)]
    cam = cams_host[0]
    log(f"[rank {rank}] scene P={P} C={C} {W}x{H} generated in {time.time() - t0:.1f}s")
    s = scene.to(dev)
    cams = [cm.to(dev) for cm in cams_host]
    empty = torch.Tensor([])
    raster.set_blend_variant(args.variant)

    # inference: state buffers stay resident (as under torch.no_grad); one pool and stream per view in flight
    pools = [raster.ScratchPool() for _ in range(V)]
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(V - 1)]

    def render(i):
        c = cams[i]
        with torch.cuda.stream(streams[i]):
            return raster.rasterize_forward(
                s.bg, s.means3D, s.features, s.opacities, s.scales, s.rotations, 1.0, empty,
                c.world_view_transform, c.full_proj_transform, c.tanfovx, c.tanfovy, H, W, empty, 0,
                c.camera_center, False, False, C, False, pool=pools[i])

    def step():   # one batch: V views of the scene, all in flight together
        return [render(i) for i in range(V)]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # warm-up, first half: ONE view in flight with per-stage hipEvents -- these are the clean per-kernel
    # durations the roofline is quoted on (with several views in flight a stream's events also count the
    # time its kernels queue behind or share the GPU with the other view's).
    n_single = max(2, args.warmup // 2)
    for _ in range(2):
        render(0)
    torch.cuda.synchronize(dev)
    raster.get_stage_ms()            # drop anything parked by earlier calls
    raster.set_stage_timing(2)       # deferred hipEvent timing of the stages, no extra syncs
    for _ in range(n_single):
        render(0)
    torch.cuda.synchronize(dev)
    raster.set_stage_timing(0)
    stage_ms = raster.get_stage_ms()
    # integrity reference: every view's num_rendered, rendered alone (concurrent forwards must reproduce it)
    ref_n = []
    for i in range(V):
        ref_n.append(render(i)[0])
        torch.cuda.synchronize(dev)
    # warm-up, second half: the batch as it is timed
    for _ in range(max(2, args.warmup - n_single)):
        out = step()
    raster.set_stage_timing(2)
    barrier()
    step_marks = []
    t0 = time.perf_counter()
    mismatches = 0
    for _ in range(args.steps):
        out = step()
        mismatches += sum(int(o[0] != n) for o, n in zip(out, ref_n))   # host ints, no device work
        step_marks.append(time.perf_counter())
    barrier()
    t = time.perf_counter() - t0
    if rank == 0:   # host-side enqueue cadence (diagnostic only; the metric uses t / steps)
        prev = t0
        log("per-step host ms: " + " ".join(f"{(m - prev) * 1e3:.2f}" for prev, m in zip([t0] + step_marks[:-1], step_marks)))
        log(f"torch reserved {torch.cuda.memory_reserved(dev) / 1e9:.2f} GB")
    raster.set_stage_timing(0)
    stage_ms_timed = raster.get_stage_ms()   # per-stream event spacing inside the timed region
    out = out[0]
    if world > 1:
        tt = torch.tensor([t], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t = float(tt.item())
    ms_per_step = t / args.steps * 1e3

    # workload statistics of this rank's view (every run prints them: bytes depend on them)
    num_rendered, color, radii, geom, binn, img, _ = out
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
    bytes_blend = 4 * C * H * W + (4 * C + 28) * sum_neff + 8 * H * W + 8 * tiles
    bytes_front = 44 * P + 36 * p_vis + 36 * num_rendered
    blend_ms = stage_ms[5] + stage_ms[6]   # weights pre-pass + accumulate (one kernel each on the default path)
    achieved = bytes_blend / (blend_ms * 1e-3) if blend_ms > 0 else 0.0

    if rank == 0:
        log(f"P_vis={p_vis} L={num_rendered} tile-list mean/max={lens.mean().item():.1f}/"
            f"{int(lens.max().item())} sum_n_t_eff={sum_neff} (mean {sum_neff / tiles:.1f}/tile) "
            f"sum_n_contrib={contributors}")
        log("stage ms (mean over timed steps): " + ", ".join(
            f"{n}={v:.3f}" for n, v in zip(
                ["preprocess", "scan+readback", "duplicate", "sort", "ranges", "blend_weights", "blend_accum"], stage_ms)))
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
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.config}: P={P} Gaussians, C={C}, {H}x{W} forward render "
                                   f"(BASELINE.md config 3 generator, seed 0)",
                       "views_per_step_per_gpu": V, "hip_streams_per_gpu": V,
                       "parallelism": f"views x{world * V}: {V} in flight per GPU on {V} HIP streams, {world} GPU(s), "
                                      f"scene replicated, no collective",
                       "blend_variant": args.variant,
                       "blend_arithmetic": ("fp32 MFMA, bit-exact" if args.variant == 15 else
                                            "split-bf16 x3 MFMA products, fp32 accumulate (<= 5e-5 of the absolute "
                                            "composite; SGS_BLEND_EXACT=1 selects the bit-exact fp32 MFMA path)")},
            # the forward blend = blend_weights_kernel + blend_accum_sweep_kernel (one launch each);
            # SURVEY 8(d)'s algorithmic bytes are a property of the pair, so the roofline is quoted
            # on the pair; the per-kernel live durations are alongside (rocprof: profiles/).
            "roofline": {"bound": "hbm", "kernel": "blend_fwd (blend_weights_kernel + blend_accum_sweep_kernel)",
                         "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": traffic, "algorithmic_bytes": bytes_blend,
                         "kernel_ms": blend_ms,
                         "kernels_ms": {"blend_weights": round(stage_ms[5], 4), "blend_accum": round(stage_ms[6], 4)},
                         "measured": f"hipEvents on the launch stream, {n_single} warm-up frames with one view in flight; "
                                     f"the timed region keeps {V} in flight (stage_ms_timed_region)"},
            "ms_per_view": ms_per_step / V,
            # bytes_alg of the WHOLE forward (blend + binning front end) over the frame time
            "whole_forward": {"algorithmic_bytes": bytes_blend + bytes_front,
                              "achieved_GBps": V * (bytes_blend + bytes_front) / (ms_per_step * 1e-3) / 1e9,
                              "frac_of_hbm_peak": V * (bytes_blend + bytes_front) / (ms_per_step * 1e-3) / HBM_PEAK},
            "stage_ms": dict(zip(["preprocess", "scan_readback", "duplicate", "sort", "ranges", "blend_weights",
                                  "blend_accum"],
                                 [round(v, 4) for v in stage_ms])),
            "stage_ms_timed_region": dict(zip(["preprocess", "scan_readback", "duplicate", "sort", "ranges",
                                               "blend_weights", "blend_accum"], [round(v, 4) for v in stage_ms_timed])),
            "integrity": {"forwards_checked": args.steps * V, "num_rendered_mismatches_vs_serial": mismatches},
            "workload_stats": {"P_vis": p_vis, "num_rendered": num_rendered, "sum_n_t_eff": sum_neff,
                               "tiles": tiles},
        }
        if world == 1 and not args.no_cpu_baseline:
            del out, color
            res["cpu_baseline"] = cpu_baseline(scene, cam, C, W, H)
            log("cpu_baseline: " + res["cpu_baseline"]["sample"])
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

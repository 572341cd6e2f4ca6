"""One view of a Gaussian-sharded scene over RCCL (launched by torch.distributed.run with >= 2 ranks, one GPU each):
rank r holds depth slab r, renders its (A, T) partial with the HIP rasteriser, dist.render_gaussian_sharded exchanges the
bands point to point and composites them with sgs_composite_over; rank 0 checks the result against its own single
render of the whole scene.  Prints GAUSSIAN_SHARDED_OK on success.   usage: torchrun --nproc-per-node N this.py [P] [C]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-gaussians_amd"))
import torch
import torch.distributed as dist


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    # SGS_TEST_ONE_DEVICE=1 (test hook): every rank on cuda:0 and gloo instead of RCCL (which refuses two ranks per device) --
    # the same code path (HIP partials, band exchange as grouped point-to-point operations, HIP composite) on a 1-GPU box
    one_dev = os.environ.get("SGS_TEST_ONE_DEVICE", "0") == "1"
    dev = torch.device("cuda", 0 if one_dev else local)
    torch.cuda.set_device(dev)
    if one_dev:
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=dev)
    from sgs_hip import raster, dist as sdist
    from sgs_hip.synthetic import make_scene
    from sgs_hip.camera import pinhole
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
    C = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    W, H, fx = 1296, 968, 1170.0
    scene = make_scene(P, C, W, H, fx, seed=11)        # every rank generates the same scene and keeps its slab
    cam = pinhole(W, H, fx).to(dev)
    order = torch.argsort(scene.means3D[:, 2])
    lo, hi = rank * P // world, (rank + 1) * P // world
    idx = order[lo:hi]
    sh = [t[idx].to(dev) for t in (scene.means3D, scene.features, scene.opacities, scene.scales, scene.rotations)]
    bg = torch.linspace(0.0, 1.0, C, device=dev)

    def partial():
        A, T, _ = raster.render_partial(sh[0], sh[1], sh[2], sh[3], sh[4], cam.world_view_transform, cam.full_proj_transform,
                                        cam.tanfovx, cam.tanfovy, H, W, cam.camera_center, bands=world if C % 128 == 0 else 0)
        return A, T
    full = sdist.render_gaussian_sharded(partial, bg, all_gather=True)
    torch.cuda.synchronize(dev)
    ok = True
    if rank == 0:
        s = scene.to(dev)
        e = torch.Tensor([])
        n, whole, _, _, _, img, _ = raster.rasterize_forward(bg, s.means3D, s.features, s.opacities, s.scales, s.rotations, 1.0, e,
                                                             cam.world_view_transform, cam.full_proj_transform, cam.tanfovx,
                                                             cam.tanfovy, H, W, e, 0, cam.camera_center, False, False, C, False)
        T_whole = raster.image_views(img, W, H)["final_T"]
        scale = float(whole.abs().max())
        # The two renders may only differ beyond fp32 rounding where the SINGLE render took the reference's T < 1e-4 stop
        # (it acts per shard in the sharded render): such a pixel ends with 1e-4 <= T < 1e-2 (alpha <= 0.99), and what lies
        # behind the stop is at most T * (|bg| + max |f|).  Everywhere else (T >= 1e-2: no stop in either render) the
        # composite must agree to rounding -- a wrong shard order or operator there must not hide in an O(1) slack.
        slack = torch.where(T_whole < 1e-2, T_whole * (float(bg.abs().max()) + float(s.features.abs().max())) * 1.001,
                            torch.zeros_like(T_whole))[None] + 1e-5 * scale
        ok = full.shape == whole.shape and bool(((full - whole).abs() <= slack).all())
        print(f"world {world}: max |sharded - single| = {float((full - whole).abs().max()):.3e} (scale {scale:.3f})", flush=True)
    flag = torch.tensor([1 if ok else 0], device="cpu" if one_dev else dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0 and int(flag.item()) == 1:
        print("GAUSSIAN_SHARDED_OK", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()

"""Does rendering two independent views on two streams from one host thread overlap usefully?
Prints ms per view for 1 and 2 streams (cfg3).  Development experiment."""
import os, sys, time
os.environ.setdefault("PYTORCH_HIP_ALLOC_CONF", "max_split_size_mb:256")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-gaussians_amd"))
sys.path.insert(0, ROOT)
import torch
from sgs_hip import raster
from sgs_hip.synthetic import CONFIGS, make_scene
from bench import view_camera

dev = torch.device("cuda", 0)
P, C, W, H, fx = CONFIGS["cfg3"]
scene = make_scene(P, C, W, H, fx, seed=0).to(dev)
cams = [view_camera(i, W, H, fx).to(dev) for i in range(2)]
empty = torch.Tensor([])
for nstreams in [int(a) for a in sys.argv[1:]] or [1, 2]:
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    pools = [raster.ScratchPool() for _ in range(nstreams)]

    def step(i):
        c = cams[i % 2]
        with torch.cuda.stream(streams[i % nstreams]):
            return raster.rasterize_forward(scene.bg, scene.means3D, scene.features, scene.opacities, scene.scales,
                                            scene.rotations, 1.0, empty, c.world_view_transform, c.full_proj_transform,
                                            c.tanfovx, c.tanfovy, H, W, empty, 0, c.camera_center, False, False, C, False,
                                            pool=pools[i % nstreams])
    for i in range(8):
        out = step(i)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    N = 40
    for i in range(N):
        out = step(i)
    torch.cuda.synchronize(dev)
    print(f"{nstreams} stream(s): {(time.perf_counter() - t0) / N * 1e3:.3f} ms per view")
    del pools, out

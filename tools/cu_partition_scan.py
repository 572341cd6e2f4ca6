"""Round 6, VERDICT r5 item 1: views pipelined on CU-masked streams.  cfg3 (1M x 512 x 968x1296), V views in flight, the host pattern of the
bench's headline (raster.rasterize_forward_inference); for every (front CUs, V): ms per view, Gpx.ch/s and the per-stream stage times inside
the region.  front = 0 is round 5's arrangement (ordinary streams, every kernel anywhere).  Every forward's num_rendered is compared with
the serial render.   usage: cu_partition_scan.py [steps=150] [fronts=0,16,32,48,64] [views=2,3,4]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-gaussians_amd")):
    sys.path.insert(0, p)
os.environ.setdefault("PYTORCH_HIP_ALLOC_CONF", "max_split_size_mb:256")
import torch
from bench import view_camera, STAGES
from sgs_hip import raster
from sgs_hip.synthetic import CONFIGS, make_scene

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 150
FRONTS = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,16,32,48,64").split(",")]
VIEWS = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "2,3,4").split(",")]
DEV = torch.device("cuda", 0)
torch.cuda.set_device(DEV)
P, C, W, H, fx = CONFIGS["cfg3"]
s = make_scene(P, C, W, H, fx, seed=0).to(DEV)
NCAM = 8
e = torch.Tensor([])
print(f"{torch.cuda.get_device_name(0)}: {raster.x16_cu_ownership()=} steps {STEPS}", flush=True)


def run(front, V):
    cams = [[view_camera(i * NCAM + k, W, H, fx).to(DEV) for k in range(NCAM)] for i in range(V)]
    part = raster.PartitionedStreams(DEV, front, V) if front > 0 else None
    streams = part.streams if part else [torch.cuda.Stream(DEV) for _ in range(V)]
    pools = [raster.ScratchPool() for _ in range(V)]

    def render(i, k, fn=raster.rasterize_forward_inference):
        c = cams[i][k % NCAM]
        with torch.cuda.stream(streams[i]):
            return fn(s.bg, s.means3D, s.features, s.opacities, s.scales, s.rotations, 1.0, e, c.world_view_transform, c.full_proj_transform,
                      c.tanfovx, c.tanfovy, H, W, e, 0, c.camera_center, False, False, C, False, pool=pools[i])
    for st in streams:
        st.wait_stream(torch.cuda.current_stream(DEV))
    ref = [[None] * NCAM for _ in range(V)]
    for i in range(V):
        for k in range(NCAM):
            ref[i][k] = render(i, k, raster.rasterize_forward)[0]
            torch.cuda.synchronize()
    out = [None] * V
    for k in range(12):
        for i in range(V):
            out[i] = None
            out[i] = render(i, k)
    torch.cuda.synchronize()
    raster.get_stage_ms()
    raster.set_stage_timing(2)
    bad = 0
    t0 = time.perf_counter()
    for k in range(STEPS):
        for i in range(V):
            out[i] = None
            out[i] = render(i, k)
            bad += int(out[i][0] != ref[i][k % NCAM])
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    raster.set_stage_timing(0)
    stg = raster.get_stage_ms()
    ms_view = t / (STEPS * V) * 1e3
    print(f"front {front:3d} CUs  views {V}: {ms_view:.4f} ms/view  {H * W * C / (ms_view * 1e-3) / 1e9:7.1f} Gpx.ch/s  mismatches {bad}  "
          + " ".join(f"{n}={v:.3f}" for n, v in zip(STAGES, stg)), flush=True)
    del out, pools
    if part:
        part.close()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()


for V in VIEWS:
    for f in FRONTS:
        run(f, V)
# one view alone on the partitions: what the sweep loses with fewer compute units (the "burst is the chip's write rate" question)
for f in FRONTS:
    run(f, 1)

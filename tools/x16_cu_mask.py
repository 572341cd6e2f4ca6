"""x16 bisect, cut (ii) in a sharper form: does the damage need the victim and the aggressor on the SAME compute units?
Views of the small C = 128 scene are rendered NF (default 4) in flight on as many HIP streams, so that view k + 1's preprocess (the kernel every event
was ever seen in) runs beside view k's x16 sweep, and every view's num_rendered / radii are compared with the serial render
(tools/repro_x16_neighbour_corruption.py's protocol).  Two legs per variant, alternating:
  shared    both streams are ordinary streams: the two kernels can land on the same CUs / SIMDs
  disjoint  the streams are created with hipExtStreamCreateWithCUMask, stream k with bits [k n / NF, (k + 1) n / NF) of the CU mask: a view's kernels only ever
            run on its stream's share of the chip's CUs, views in flight together on different shares -- victim and aggressor never share a CU.
Events under `shared` and none under `disjoint` = a same-CU resource; events under both = something chip-wide (power, clocks, fabric).
Needs a library built with `make X16=1 EXPERIMENTS=1`.   usage: x16_cu_mask.py [rounds=3000] [variant=0x6F] [in_flight=4] [x8 control=1] [backward too=0]"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "semantic-gaussians_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from helpers import small_scene
from sgs_hip import raster
from sgs_hip.camera import pinhole

DEV = "cuda:0"
R = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
VAR = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0x6F
NF = int(sys.argv[3]) if len(sys.argv) > 3 else 4          # views in flight = streams (the events of rounds 2-4 were seen with four)
CONTROL = (sys.argv[4] != "0") if len(sys.argv) > 4 else True
BWD = (sys.argv[5] != "0") if len(sys.argv) > 5 else False   # every view also runs its backward (the fused kernel issues x16 products too) on its stream
scene, _ = small_scene(P=5000, C=128, W=208, H=128, fx=170.0, seed=5)
s = scene.to(DEV)
cams = [pinhole(208, 128, fx).to(DEV) for fx in (150.0, 160.0, 170.0, 180.0, 190.0, 200.0)]
e = torch.Tensor([])
pools = [raster.ScratchPool() for _ in range(NF)]
hip = C.CDLL("libamdhip64.so")
ncu = torch.cuda.get_device_properties(0).multi_processor_count
words = (ncu + 31) // 32


def masked_stream(lo, hi):
    m = (C.c_uint32 * words)()
    for b in range(lo, hi):
        m[b // 32] |= 1 << (b % 32)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(words), m)
    assert rc == 0, f"hipExtStreamCreateWithCUMask: {rc}"
    return torch.cuda.ExternalStream(st.value, device=DEV)


dL = torch.randn(128, 128, 208, device=DEV)


def render(c, slot):
    out = raster.rasterize_forward(s.bg, s.means3D, s.features, s.opacities, s.scales, s.rotations, 1.0, e, c.world_view_transform,
                                   c.full_proj_transform, c.tanfovx, c.tanfovy, 128, 208, e, 0, c.camera_center, False, False, 128, False,
                                   pool=None if BWD else pools[slot])
    if BWD:
        raster.rasterize_backward(s.bg, s.means3D, out[2], s.features, s.scales, s.rotations, 1.0, e, c.world_view_transform,
                                  c.full_proj_transform, c.tanfovx, c.tanfovy, dL, e, 0, c.camera_center, out[3], out[0], out[4], out[5], False)
    return out[0], out[2].clone()


def leg(name, streams, rounds, serial):
    bad = 0
    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
    for _ in range(rounds):
        outs = []
        for i, c in enumerate(cams):
            with torch.cuda.stream(streams[i % NF]):
                outs.append(render(c, i % NF))
        torch.cuda.synchronize()
        for vi, (a, b) in enumerate(zip(serial, outs)):
            wrong = a[0] != b[0] or not torch.equal(a[1], b[1])
            if wrong and bad < 4:
                d = torch.nonzero(a[1] != b[1]).flatten()
                print(f"  [{name}] view {vi}: num_rendered {a[0]} vs {b[0]}; {d.numel()} radii differ at {d[:12].tolist()}", flush=True)
            bad += int(wrong)
    print(f"{name}: {bad} corrupted forwards of {rounds * len(cams)}", flush=True)
    return bad


print(f"{torch.cuda.get_device_name(0)}: {ncu} CUs, mask words {words}; build flags {raster.build_flags()}; variant {VAR:#x}, {R} rounds per leg", flush=True)
shared = [torch.cuda.Stream(DEV) for _ in range(NF)]
disjoint = [masked_stream(k * ncu // NF, (k + 1) * ncu // NF) for k in range(NF)]   # stream k owns CUs [k n / NF, (k + 1) n / NF)
legs = ((VAR, f"x16 {VAR:#x}"), (0x6E if raster.build_flags() & 4 else 0, "x8 control")) if CONTROL else ((VAR, f"x16 {VAR:#x}"),)
for variant, label in legs:
    raster.set_blend_variant(variant)
    serial = [render(c, 0) for c in cams]
    torch.cuda.synchronize()
    tot = {"shared": 0, "disjoint": 0}
    for rep in range(2):
        tot["shared"] += leg(f"{label} shared CUs #{rep}", shared, R // 2, serial)
        tot["disjoint"] += leg(f"{label} disjoint CU halves #{rep}", disjoint, R // 2, serial)
    print(f"== {label}: shared {tot['shared']} / disjoint {tot['disjoint']} corrupted of {R * len(cams)} each", flush=True)
raster.set_blend_variant(0)

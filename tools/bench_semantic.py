"""cfg3 geometry (1M Gaussians, C=512, 968x1296): semantic labels per view, the reference's consumer
(render the feature map, normalise, einsum, argmax) vs the projected render (sgs_hip.semantic)."""
import os, sys, time
os.environ.setdefault("PYTORCH_HIP_ALLOC_CONF", "max_split_size_mb:256")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-gaussians_amd"))
import torch
import channel_rasterization as cr
from sgs_hip import semantic
from sgs_hip.synthetic import CONFIGS, make_scene
from sgs_hip.camera import pinhole

dev = "cuda:0"
P, C, W, H, fx = CONFIGS["cfg3"]
scene = make_scene(P, C, W, H, fx, seed=0).to(dev)
cam = pinhole(W, H, fx).to(dev)
for n_cls in (21, 161):
    g = torch.Generator().manual_seed(1)
    text = torch.randn(n_cls, C, generator=g)
    text = (text / text.norm(dim=1, keepdim=True)).to(dev)
    settings = cr.GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=scene.bg, scale_modifier=1.0,
        viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=0, campos=cam.camera_center,
        prefiltered=False, debug=False, num_channels=C)
    a = (scene.means3D, scene.opacities, scene.scales, scene.rotations)

    def reference_way():
        sim = semantic.render_similarity(settings, *a, scene.features, text, normalised=True)
        return sim[1:].argmax(dim=0)

    proj = semantic.project_features(scene.features, text)

    def fast_way():
        return semantic.labels_from_logits(semantic.render_logits(settings, *a, proj, text)[0])

    with torch.no_grad():
        for fn in (reference_way, fast_way):
            for _ in range(3):
                lab = fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                lab = fn()
            torch.cuda.synchronize()
            print(f"n_cls={n_cls:4d} {fn.__name__:14s} {(time.perf_counter() - t0) / 10 * 1e3:7.2f} ms per view")
        agree = (reference_way() == fast_way()).float().mean().item()
        print(f"n_cls={n_cls:4d} label agreement {agree:.5f}")

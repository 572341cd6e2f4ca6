"""N3 measurement (lives under tests/ because its host leg uses the oracle): one view of the fusion step at fusion_scannet.yaml's image size (648x484), 1M Gaussian
centres, a 512-channel 2-D feature map, occlusion test against a rendered depth map.

  device : sgs_hip.fusion (mapping kernel + accumulate kernel; (C,H,W) -> (H,W,C) transpose included)
  host   : the reference's data flow (fusion.py:127-147) with the NumPy oracle standing in for the
           reference class: centres + depth copied to the host, NumPy mapping, host gather of
           features[:, y, x], (N, C) block copied back, masked += on the device."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-gaussians_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle import fusion_oracle as fo
from sgs_hip.fusion import PointCloudToImageMapper, accumulate_features

dev = "cuda:0"
N, W, H, C = 1_000_000, 648, 484, 512
rng = np.random.default_rng(0)
intr = np.array([[577.0, 0, 319.5], [0, 577.0, 239.5], [0, 0, 1.0]])
w2c = np.eye(4); w2c[2, 3] = 3.0
wvt = torch.from_numpy(w2c.T.astype(np.float32)).to(dev)
xyz = torch.from_numpy((rng.normal(size=(N, 3)) * np.array([2.0, 1.5, 1.0])).astype(np.float32)).to(dev)
depth = torch.from_numpy(rng.uniform(2.0, 4.0, size=(H, W)).astype(np.float32)).to(dev)
features = torch.randn(C, H, W, device=dev)
mapper = PointCloudToImageMapper((W, H), intrinsics=intr, cut_bound=10, device=dev)
feat_sum = torch.zeros(N, C, device=dev); times = torch.zeros(N, device=dev)

def device_view():
    mapping, _ = mapper.compute_mapping_device(wvt, xyz, depth)
    accumulate_features(feat_sum, times, features, mapping)
    return mapping

for _ in range(3): m = device_view()
torch.cuda.synchronize()
t0 = time.perf_counter(); R = 10
for _ in range(R): m = device_view()
torch.cuda.synchronize()
t_dev = (time.perf_counter() - t0) / R
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
ev[0].record(); mapping, _ = mapper.compute_mapping_device(wvt, xyz, depth); ev[1].record()
f_hwc = features.permute(1, 2, 0).contiguous(); ev[2].record()
accumulate_features(feat_sum, times, f_hwc, mapping, channel_last=True); ev[3].record()
torch.cuda.synchronize()
nvis = int((m[:, 2] != 0).sum())

def host_view():
    mp, _ = fo.compute_mapping(wvt.cpu().numpy(), xyz.cpu().numpy(), (W, H), mapper.intrinsics, 10, 0.25, depth.cpu().numpy())
    mp = torch.from_numpy(mp)
    fm = features.cpu()[:, mp[:, 0], mp[:, 1]].permute(1, 0).to(dev)
    mask_k = (mp[:, 2] != 0).to(dev)
    times[mask_k] += 1
    feat_sum[mask_k] += fm[mask_k]

host_view(); torch.cuda.synchronize()
t0 = time.perf_counter(); host_view(); torch.cuda.synchronize()
t_host = time.perf_counter() - t0
print(f"fusion step, N={N} points ({nvis} visible), {W}x{H}, C={C}: device {t_dev * 1e3:.2f} ms per view "
      f"(mapping kernel {ev[0].elapsed_time(ev[1]):.3f} ms, (C,H,W)->(H,W,C) transpose {ev[1].elapsed_time(ev[2]):.3f} ms, "
      f"accumulate kernel {ev[2].elapsed_time(ev[3]):.3f} ms); reference data flow through the host {t_host * 1e3:.0f} ms per view")

"""The pure-PyTorch CPU splat (oracle/torch_splat.py: the CPU baseline BASELINE.json names, cfg1's renderer)
against the C oracle on the same inputs: same lists, same n_contrib, feature map within fp32 rounding of a
different summation order (matmul vs sequential fma)."""
import numpy as np
import torch

from helpers import small_scene, oracle_forward
from oracle import torch_splat


def test_torch_splat_matches_oracle(orc):
    scene, cam = small_scene(P=3000, C=6, W=112, H=80, fx=100.0, seed=3)
    scene = scene._replace(bg=torch.linspace(-0.5, 0.5, 6))
    fw = oracle_forward(orc, scene, cam)
    r = torch_splat.render(scene, cam, 112, 80)
    assert r["num_rendered"] == fw["num_rendered"]
    assert np.array_equal(r["radii"].numpy(), fw["radii"])
    assert np.array_equal(r["point_list"].numpy().astype(np.uint32), fw["point_list"])
    assert np.array_equal(r["ranges"].numpy().astype(np.uint32), fw["ranges"])
    # decisions at the alpha / transmittance thresholds use libm exp here and the contract exp in the oracle:
    # allow a handful of pixels to differ by one list entry
    nc = r["n_contrib"].numpy()
    assert (nc != fw["n_contrib"]).mean() < 2e-3
    err = np.abs(r["out"].numpy() - fw["out"])
    assert np.quantile(err, 0.999) < 2e-6 and err.max() < 5e-3
    assert np.abs(r["final_T"].numpy() - fw["final_T"]).max() < 5e-3


def test_cfg1_renders_on_the_cpu(orc):
    """BASELINE config 1: 10k Gaussians, 256x256, C = 3 -- the pure-PyTorch CPU splat, checked against the oracle."""
    from sgs_hip.synthetic import make_config
    scene, cam = make_config("cfg1")
    t = {}
    r = torch_splat.render(scene, cam, 256, 256, timings=t)
    fw = oracle_forward(orc, scene, cam)
    assert r["num_rendered"] == fw["num_rendered"] and np.array_equal(r["radii"].numpy(), fw["radii"])
    assert np.array_equal(r["point_list"].numpy().astype(np.uint32), fw["point_list"])
    err = np.abs(r["out"].numpy() - fw["out"])
    assert np.quantile(err, 0.999) < 2e-6
    assert t["walked"] > 0 and t["blend_s"] > 0


def test_tile_subset_matches_full_render():
    scene, cam = small_scene(P=1500, C=4, W=64, H=48, fx=60.0, seed=1)
    full = torch_splat.render(scene, cam, 64, 48)
    part = torch_splat.render(scene, cam, 64, 48, tile_ids=[5, 6])
    assert torch.equal(part["out"][:, 16:32, 16:48], full["out"][:, 16:32, 16:48])

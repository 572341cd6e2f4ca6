"""Round 3: the accumulate sweep of blend_sweep2.hip (LDS-polled DMA arrival, stores spread over the next tile) against
the oracle, through the C-ABI.  Variants (low nibble of the blend variant; 0x60 = 48-tile segments, the default):
  0x6B  exact fp32 MFMA   -- bit-identical feature map;
  0x6A  f32-equivalent    -- six bf16 products of the exact three-term splits; |error| <= X6_TOL * sum |f| w;
  0x6E  the same with the weights pre-split by the weights pre-pass (round 3's default);
  0x66  round 4's default: the 0x6E arithmetic in the ping-pong sweep (one 8-wave workgroup for both row parities,
        MFMA phase of one half beside the load / store phase of the other) -- bit-identical to 0x6E;
  0x10066  round 5's default: the ping-pong sweep on the double-rate v_mfma_f32_32x32x16_bf16 (the same six products, 16 k per
        instruction instead of 8: a feature map of its own, held to the same bounds);
  (0x6C / 0x6F, the double-rate-MFMA experiments, are not in the product library: make X16=1.)
Every integer output stays bit-exact (it comes from the shared front end and weights pre-pass)."""
import numpy as np
import pytest
import torch

from helpers import small_scene, oracle_forward, need_experiments
from test_parity_gpu import _hip_forward

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

V_X6, V_EXACT, V_X6W, V_X6S, V_X6P, V_X6PW = 0x6A, 0x6B, 0x6C, 0x6D, 0x6E, 0x6F
V_PP = 0x66       # round 4: ping-pong sweep on the x8 MFMA (0x6_ pins 48-tile segments like the others)
V_PP16 = 0x10066  # round 5: the ping-pong sweep on the double-rate v_mfma_f32_32x32x16_bf16 (same products, 16 k per instruction), lock step
V_FR16 = 0x10064  # round 5's default kernel: the same with the halves free-running on per-stage LDS counters instead of two barriers per step (bit-identical to V_PP16)
V_FR16_S1 = 0x110064   # round 6, THE DEFAULT kernel: V_FR16 with store placement 1 (bits [21:20] of the word): pixel block 1 of a finished pair leaves in the next tile's first matrix phase
V_X6C = 0x67      # six products, fp32 weights handed over, split once per workgroup into LDS: bit-identical to V_X6P
SHIPS = (0, 14, 15, V_EXACT, V_PP, V_PP16, V_FR16, V_FR16_S1)   # everything else is a development form (make EXPERIMENTS=1): its tests skip on the product library


def _gate(variant):
    if variant not in SHIPS:
        need_experiments(f"blend variant {variant:#x}")
# Against the fp32 ORACLE the difference is dominated by the oracle's own roundings: its multiply-add chain rounds once
# per contribution (<= 2^-24 |partial sum| each, K ~ 50-300 contributions), the six-product path drops
# F2 W3 + F3 W2 + F3 W3 <= 2^-23 |f w| per term and rounds once per MFMA.  4e-6 of the ABSOLUTE composite
# (sum |f| w + T |bg|) covers both and is 25x inside the north star's 1e-4.  The sharper statement is
# check_vs_f64 below: against the exact (float64) composite the path is as accurate as the oracle's fp32 chain.
X6_TOL = 4e-6


def check(orc, scene, cam, variant, seg=None, **kw):
    v = variant if seg is None else (variant & ~0xF0) | (seg << 4)   # (the segment nibble only: the word's other fields stay)
    fw = oracle_forward(orc, scene, cam, **kw)
    n, color, radii, geom, binn, img, depth = _hip_forward(scene, cam, variant=v, **kw)
    from sgs_hip import raster
    W, H = cam.image_width, cam.image_height
    assert n == fw["num_rendered"]
    assert np.array_equal(radii.cpu().numpy(), fw["radii"])
    im = {k: t.cpu().numpy() for k, t in raster.image_views(img, W, H).items()}
    assert np.array_equal(im["n_contrib"].view(np.uint32), fw["n_contrib"])
    assert np.array_equal(im["final_T"].view(np.uint32), fw["final_T"].view(np.uint32))
    out = color.cpu().numpy()
    if (variant & 15) == 11:
        assert np.array_equal(out.view(np.uint32), fw["out"].view(np.uint32))
        return fw, 0.0
    sa = scene._replace(features=scene.features.abs(), bg=scene.bg.abs())
    fa = oracle_forward(orc, sa, cam, **kw)["out"]
    err = np.abs(out - fw["out"])
    assert (err <= X6_TOL * fa + 1e-30).all(), float((err / (fa + 1e-30)).max())
    # "f32-equivalent": measured against the EXACT composite (float64 sums of the same fp32 weights), the six-product
    # path is no less accurate than the reference's own fp32 multiply-add chain
    bgv = scene.bg.numpy() if kw.get("bg") is None else np.asarray(kw["bg"], np.float32)
    truth = orc.blend_forward_f64(fw, fw, fw["features"], bgv, W, H)
    e_hip = np.abs(out.astype(np.float64) - truth) / (fa + 1e-30)
    e_orc = np.abs(fw["out"].astype(np.float64) - truth) / (fa + 1e-30)
    assert e_hip.max() <= 1.5 * e_orc.max() + 2.0 ** -22, (e_hip.max(), e_orc.max())
    assert np.sqrt((e_hip ** 2).mean()) <= 1.25 * np.sqrt((e_orc ** 2).mean()) + 2.0 ** -26, (np.sqrt((e_hip ** 2).mean()), np.sqrt((e_orc ** 2).mean()))
    return fw, float(e_hip.max())


SHAPES = [(128, 200, 120), (160, 208, 70), (512, 192, 100), (256, 48, 40), (128, 16, 16), (128, 400, 64), (128, 336, 48)]


@pytest.mark.parametrize("variant", [V_EXACT, V_X6, V_X6S, V_X6P, V_X6C, V_PP, V_PP16, V_FR16, V_FR16_S1])
@pytest.mark.parametrize("C,W,H", SHAPES)
def test_sweep2_shapes(orc, variant, C, W, H):
    """W % 32 == 16 (staggered pairs: a segment starts with an unpaired right half on odd rows), W % 32 == 0, ragged W
    (guarded edge pairs), a single tile, an odd tile count (trailing unpaired left half)."""
    _gate(variant)
    scene, cam = small_scene(P=3000, C=C, W=W, H=H, fx=170.0, seed=C + W)
    check(orc, scene, cam, variant)
    check(orc, scene, cam, variant, seg=1)   # 8-tile segments: many segment ends


@pytest.mark.parametrize("variant", [V_EXACT, V_X6, V_X6P, V_X6C, V_PP, V_PP16, V_FR16, V_FR16_S1])
def test_sweep2_background_and_short_lists(orc, variant):
    """Non-zero background (the closing T * bg pseudo entry), tiles whose only entry is that pseudo entry."""
    _gate(variant)
    scene, cam = small_scene(P=60, C=128, W=208, H=96, fx=170.0, seed=5)
    g = torch.Generator().manual_seed(3)
    scene = scene._replace(bg=torch.randn(128, generator=g), scales=scene.scales * 0.3)
    fw, _ = check(orc, scene, cam, variant)
    r = fw["ranges"].reshape(-1, 2)
    assert (r[:, 0] == r[:, 1]).any()   # empty tiles exist


@pytest.mark.parametrize("variant", [V_EXACT, V_X6, V_X6P, V_X6C, V_PP, V_PP16, V_FR16, V_FR16_S1])
def test_sweep2_long_lists(orc, variant):
    """Dense scene, wide image: the batch-table window (1024 batches) slides, chunk tables run past one chunk per tile,
    deferred stores ride along tiles of very different lengths."""
    _gate(variant)
    scene, cam = small_scene(P=40000, C=128, W=784, H=32, fx=600.0, seed=77)
    scene = scene._replace(scales=scene.scales * 3.0, opacities=scene.opacities * 0.05)
    check(orc, scene, cam, variant)        # (first frame may take the overflow fallback)
    fw, _ = check(orc, scene, cam, variant)
    assert fw["n_contrib"].max() > 900
    check(orc, scene, cam, variant, seg=6)   # one 49-tile segment: ~2900 batches


@pytest.mark.parametrize("variant", [V_EXACT, V_X6, V_X6P, V_X6C, V_PP, V_PP16, V_FR16, V_FR16_S1])
def test_sweep2_padded_pitch(orc, variant):
    """Rows padded to 32 pixels (SGS_OPT_OUT_PITCH): every pair is interior, no stagger."""
    _gate(variant)
    from sgs_hip import raster
    scene, cam = small_scene(P=2500, C=128, W=203, H=90, fx=170.0, seed=9)
    raster.OUTPUT_PITCH_ALIGN = 32
    try:
        check(orc, scene, cam, variant)
    finally:
        raster.OUTPUT_PITCH_ALIGN = 0


@pytest.mark.parametrize("V", [V_X6P, V_PP, V_PP16, V_FR16, V_FR16_S1])
def test_sweep2_deterministic_under_load(orc, V):
    """The same frame 300 times with two other views in flight on other streams: every feature map bit-identical
    (a stale ring stage -- a bundle consumed before it landed -- would show up as a differing map)."""
    _gate(V)
    from sgs_hip import raster
    scene, cam = small_scene(P=6000, C=256, W=400, H=160, fx=300.0, seed=21)
    ref = _hip_forward(scene, cam, variant=V)[1].clone()
    side = [torch.cuda.Stream(device=DEV) for _ in range(2)]
    bad = 0
    for it in range(300):
        for st in side:
            with torch.cuda.stream(st):
                _hip_forward(scene, cam, variant=V)
        out = _hip_forward(scene, cam, variant=V)[1]
        bad += int(not torch.equal(out, ref))
    torch.cuda.synchronize()
    assert bad == 0


@pytest.mark.parametrize("variant", [0, V_EXACT, V_X6P, V_PP, V_PP16, V_FR16, 14])
def test_feature_scale_invariance_is_bit_exact(variant):
    """A size-independent property of every accumulate arithmetic: scaling the features and the background by a power of two
    scales the feature map by exactly that power of two (roundings commute with 2^k away from over / underflow) -- for the
    fp32 chain, for the three-term bf16 splits (split3(2^k x) = 2^k split3(x)) and for the two-term split.  A term that
    lost bits on the way (a non-exact split, a flushed residual) would break the identity."""
    _gate(variant)
    scene, cam = small_scene(P=5000, C=256, W=208, H=96, fx=170.0, seed=91)
    g = torch.Generator().manual_seed(4)
    scene = scene._replace(bg=torch.randn(256, generator=g))
    base = _hip_forward(scene, cam, variant=variant)[1]
    for k in (-60, -7, 9, 70):
        s2 = scene._replace(features=scene.features * 2.0 ** k, bg=scene.bg * 2.0 ** k)
        out = _hip_forward(s2, cam, variant=variant)[1]
        assert torch.equal(out, base * 2.0 ** k), k


def test_x16_sweep_forms_agree_bitwise():
    """The default (free-running halves on the x16 MFMA, its own segment length), the same with pinned segment lengths and the lock-step x16
    form issue the same products in the same order into the same accumulators: bit-identical maps -- staggered and plain pitches, several
    channel chunks, a non-zero background, short segments (8 tiles: many unpaired halves) and long lists (the table window slides)."""
    cases = [(4000, 256, 208, 96, 170.0, 1, 1.0), (30000, 128, 400, 64, 170.0, 2, 1.0), (500, 512, 48, 40, 170.0, 3, 1.0),
             (3000, 128, 336, 48, 170.0, 4, 1.0), (40000, 128, 784, 32, 600.0, 77, 3.0)]
    for (P, C, W, H, fx, seed, sc) in cases:
        scene, cam = small_scene(P=P, C=C, W=W, H=H, fx=fx, seed=seed)
        g = torch.Generator().manual_seed(seed)
        scene = scene._replace(bg=torch.randn(C, generator=g), scales=scene.scales * sc, opacities=scene.opacities * (0.05 if sc > 1 else 1.0))
        for _ in range(2):   # (let the stream's work-list arena grow to this scene)
            _hip_forward(scene, cam, variant=0)
        d = _hip_forward(scene, cam, variant=0)[1]
        for segn in (6, 1, 3):
            assert torch.equal(d, _hip_forward(scene, cam, variant=0x10006 | (segn << 4))[1]), ("lock step", P, C, W, H, segn)
            assert torch.equal(d, _hip_forward(scene, cam, variant=0x10004 | (segn << 4))[1]), ("free running", P, C, W, H, segn)
            # round 6: where a finished pair's stores are issued (the steps then pair pixel blocks 0, 2 | 1, 3) changes no sum's order
            assert torch.equal(d, _hip_forward(scene, cam, variant=0x110004 | (segn << 4))[1]), ("store placement 1 (the default word)", P, C, W, H, segn)
            assert torch.equal(d, _hip_forward(scene, cam, variant=0x10004 | (segn << 4))[1]), ("round 5's default", P, C, W, H, segn)


def test_cooperative_split_equals_presplit_bitwise():
    """V_X6C and V_X6P evaluate the same six products of the same three-term splits (split3 in the sweep / in the weights
    pre-pass): the feature maps are bit-identical."""
    need_experiments("round 3's sweeps (0x6E / 0x67)")
    for (P, C, W, H, seed) in ((4000, 256, 208, 96, 1), (30000, 128, 400, 64, 2), (500, 512, 48, 40, 3)):
        scene, cam = small_scene(P=P, C=C, W=W, H=H, fx=170.0, seed=seed)
        g = torch.Generator().manual_seed(seed)
        scene = scene._replace(bg=torch.randn(C, generator=g))
        a = _hip_forward(scene, cam, variant=V_X6P)[1]
        b = _hip_forward(scene, cam, variant=V_X6C)[1]
        assert torch.equal(a, b)


def test_ping_pong_sweep_equals_round3_sweep_bitwise():
    """The ping-pong kernel (default) issues the same six products in the same order into the same accumulators as
    round 3's sweep: bit-identical maps, for staggered and plain pitches, several channel chunks, a non-zero background,
    short segments (seg nibble 1 = 8 tiles: many unpaired halves) and long lists (the table window slides)."""
    need_experiments("round 3's sweep and the ping-pong experiments (nibbles 14 / 5 / 4)")
    cases = [(4000, 256, 208, 96, 170.0, 1, 1.0), (30000, 128, 400, 64, 170.0, 2, 1.0), (500, 512, 48, 40, 170.0, 3, 1.0),
             (3000, 128, 336, 48, 170.0, 4, 1.0), (40000, 128, 784, 32, 600.0, 77, 3.0)]
    for (P, C, W, H, fx, seed, sc) in cases:
        scene, cam = small_scene(P=P, C=C, W=W, H=H, fx=fx, seed=seed)
        g = torch.Generator().manual_seed(seed)
        scene = scene._replace(bg=torch.randn(C, generator=g), scales=scene.scales * sc,
                               opacities=scene.opacities * (0.05 if sc > 1 else 1.0))
        for segn in (6, 1, 3):
            a = _hip_forward(scene, cam, variant=0xE | (segn << 4))[1]
            b = _hip_forward(scene, cam, variant=0x6 | (segn << 4))[1]
            assert torch.equal(a, b), (P, C, W, H, segn)
            # nibble 5 (experiment): fp32 weight rows handed over, split one step ahead inside the sweep
            c = _hip_forward(scene, cam, variant=0x5 | (segn << 4))[1]
            assert torch.equal(c, b), ("fp32 hand-over", P, C, W, H, segn)
            # nibble 4 (experiment): no barriers, the halves run free on per-stage arrival / consumption counters
            f = _hip_forward(scene, cam, variant=0x4 | (segn << 4))[1]
            assert torch.equal(f, b), ("free-running halves", P, C, W, H, segn)
        # the default is the same sweep on the x16 MFMA: a map of its own (16 k per accumulate instead of 8), the same for every segment length
        assert torch.equal(_hip_forward(scene, cam, variant=0)[1], _hip_forward(scene, cam, variant=V_PP16)[1]), (P, C, W, H)
        assert torch.equal(_hip_forward(scene, cam, variant=0)[1], _hip_forward(scene, cam, variant=V_FR16)[1]), (P, C, W, H)


def test_superbatch_weights_prepass_equals_batch16_bitwise():
    """Round 4's weights pre-pass walks a tile's list a super-batch at a time (blend_weights2_sb_kernel: two pixels per lane,
    128 entries; blend_weights_sb_kernel: lane = pixel, 256 entries; bit 15 of the variant word restores the 16-entry-batch
    kernels): same arithmetic, same entry order, same work list -- the feature map, the final transmittance and the
    contributor counts are bit-identical across all four kernels.  Cases: lists shorter than one super-batch, lists of thousands
    of entries with most of them rejected at tile level, more than 128 active entries per tile (the work list crosses chunk
    boundaries), more than 16 kept per super-batch (several groups), a non-zero background."""
    need_experiments("the superseded weights pre-passes (variant bits 14 / 15)")
    from sgs_hip import raster
    cases = [(300, 128, 64, 48, 100.0, 2, 1.0, 1.0), (6000, 256, 400, 160, 300.0, 21, 1.0, 1.0),
             (40000, 128, 784, 32, 600.0, 77, 3.0, 0.05), (60000, 128, 96, 64, 90.0, 5, 0.6, 0.08),
             (20000, 128, 48, 48, 60.0, 9, 2.0, 0.02)]
    for (P, C, W, H, fx, seed, sc, op) in cases:
        scene, cam = small_scene(P=P, C=C, W=W, H=H, fx=fx, seed=seed)
        g = torch.Generator().manual_seed(seed)
        scene = scene._replace(bg=torch.randn(C, generator=g), scales=scene.scales * sc, opacities=scene.opacities * op)
        for _ in range(3):   # (let the stream's work-list arena grow to this scene: an overflowing frame takes the exact single-kernel path)
            _hip_forward(scene, cam, variant=0x66)
        for v in (0x66, 0x6E, 0x16):
            new = _hip_forward(scene, cam, variant=v)          # two pixels per lane, 128-entry super-batches (the default)
            a = raster.image_views(new[5], W, H)
            # 0x8000: 16-entry batches, lane = pixel (round 2's kernel) | 0x4000: 256-entry super-batches, lane = pixel |
            # 0xC000: 16-entry batches, two pixels per lane (round 3's kernel)
            for alt in (0x8000, 0x4000, 0xC000):
                old = _hip_forward(scene, cam, variant=v | alt)
                assert new[0] == old[0]
                assert torch.equal(new[1], old[1]), (P, C, W, H, hex(v), hex(alt))
                b = raster.image_views(old[5], W, H)
                assert torch.equal(a["n_contrib"], b["n_contrib"]) and torch.equal(a["final_T"], b["final_T"]), (P, C, W, H, hex(v), hex(alt))


def test_x16_experiments_are_not_in_the_product_library():
    """dense v_mfma_f32_32x32x16_bf16 issue damages packed-fp32 results of FOREIGN waves resident on the same compute
    unit (round 5: 0 events with victim and aggressor on disjoint CU masks, hundreds with shared CUs).  The x16 sweeps whose
    workgroups leave room for foreign waves on their CU (round 2's, round 3's) and the filler experiments are built only by
    `make X16=1`; the default library answers their variants with an error.  (The ping-pong sweep owns its CU: its x16 form ships.)"""
    scene, cam = small_scene(P=500, C=128, W=64, H=48, fx=100.0, seed=2)
    for v in (0x6C, 0x6F, 0x16F, 0x808, 0x1F, 0x20036, 0x30066):   # (the last two: the filler forms of the x16 ping-pong sweep; its dense form is the default)
        with pytest.raises(RuntimeError, match="X16"):
            _hip_forward(scene, cam, variant=v)
    _hip_forward(scene, cam, variant=0)   # (and the stream is usable afterwards)

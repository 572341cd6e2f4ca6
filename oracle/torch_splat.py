"""Pure-PyTorch CPU splat -- the CPU baseline BASELINE.json's north_star names ("a pure-PyTorch CPU splat on the
box's own host cores"), and a second, vectorised checker.  TEST / BENCH INFRASTRUCTURE: only tests/,
bench.py's cpu_baseline leg and __graft_entry__.smoke() may import it; the product never does.

The reference has no CPU renderer (SURVEY.md finding 2: model/render_utils.py holds only palette / text helpers),
so this is the build's own: the same pipeline as the oracle written with torch ops on
whole arrays, fp32:

  preprocess   one pass of tensor algebra over all P Gaussians (projection, EWA covariance, conic, radius, rect);
  binning      one (tile << 32 | depth bits) key per (Gaussian, tile) instance, ONE stable torch.sort, ranges by
               searchsorted;
  blend        per tile, chunks of the sorted list as a [256 px x n] alpha matrix: masks for the skips, an
               exclusive cumprod for the transmittance (carried from chunk to chunk), the reference's stop rule as
               a running "dead" mask, then  acc += W @ F  -- a [256 x n] x [n x C] matmul per chunk; a tile stops
               when all its pixels are done, as the reference's block does.

Threads: whatever torch.get_num_threads() says (bench.py sets os.cpu_count()).
"""
import math
import time

import torch

TILE = 16


def preprocess(means3D, scales, rotations, opacities, view, proj, W, H, tanfovx, tanfovy, scale_modifier=1.0):
    """-> dict(depth, pix (P,2), conic (P,3), opacity (P), radii int32, rect (P,4) int64 [x0,y0,x1,y1])"""
    P = means3D.shape[0]
    f32 = torch.float32
    t = means3D @ view[:3, :3] + view[3, :3]
    hom = torch.cat([means3D, torch.ones(P, 1, dtype=f32)], 1) @ proj
    inv_w = 1.0 / (hom[:, 3] + 1e-7)
    ndc = hom[:, :2] * inv_w[:, None]
    size = torch.tensor([W, H], dtype=torch.float64)
    pix = (((ndc.double() + 1.0) * size - 1.0) * 0.5).float()      # ndc2Pix in double, rounded once
    q = rotations
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(P, 3, 3)
    M = R * (scale_modifier * scales)[:, None, :]
    S3 = M @ M.transpose(1, 2)
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    tz = t[:, 2]
    lim = torch.tensor([1.3 * tanfovx, 1.3 * tanfovy], dtype=f32)
    uv = torch.minimum(torch.maximum(t[:, :2] / tz[:, None], -lim), lim) * tz[:, None]
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -fx * uv[:, 0] / (tz * tz), zero, fy / tz, -fy * uv[:, 1] / (tz * tz)],
                    1).reshape(P, 2, 3)
    A = J @ view[:3, :3].t()
    S2 = A @ S3 @ A.transpose(1, 2)
    a, b, c = S2[:, 0, 0] + 0.3, S2[:, 0, 1], S2[:, 1, 1] + 0.3
    det = a * c - b * b
    inv = 1.0 / det
    conic = torch.stack([c * inv, -b * inv, a * inv], 1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam))
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    x0 = torch.clamp(torch.trunc((pix[:, 0] - radius) / TILE), 0, gx).long()
    x1 = torch.clamp(torch.trunc((pix[:, 0] + radius + (TILE - 1)) / TILE), 0, gx).long()
    y0 = torch.clamp(torch.trunc((pix[:, 1] - radius) / TILE), 0, gy).long()
    y1 = torch.clamp(torch.trunc((pix[:, 1] + radius + (TILE - 1)) / TILE), 0, gy).long()
    vis = (tz > 0.2) & (det != 0) & ((x1 - x0) * (y1 - y0) > 0)
    radii = torch.where(vis, radius, torch.zeros_like(radius)).to(torch.int32)
    return dict(depth=tz, pix=pix, conic=conic, opacity=opacities.reshape(-1), radii=radii,
                rect=torch.stack([x0, y0, x1, y1], 1), vis=vis)


def binning(pre, W, H):
    """-> point_list (L) int64 sorted by (tile, depth, id), ranges (tiles, 2) int64"""
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    ids = torch.nonzero(pre["vis"]).reshape(-1)
    rect = pre["rect"][ids]
    w, h = rect[:, 2] - rect[:, 0], rect[:, 3] - rect[:, 1]
    cnt = w * h
    L = int(cnt.sum())
    owner = torch.repeat_interleave(torch.arange(ids.numel()), cnt)          # instance -> visible index
    first = torch.cumsum(cnt, 0) - cnt
    k = torch.arange(L) - first[owner]                                        # index inside the rect, row-major
    tx = rect[owner, 0] + k % w[owner]
    ty = rect[owner, 1] + k // w[owner]
    depth_bits = pre["depth"][ids].view(torch.int32).long()[owner]            # depth > 0.2: bit pattern is monotone
    key = ((ty * gx + tx) << 32) | depth_bits
    order = torch.sort(key, stable=True).indices                              # ties keep ascending Gaussian id
    key_s = key[order]
    point_list = ids[owner[order]]
    tiles = torch.arange(gx * gy)
    lo = torch.searchsorted(key_s, tiles << 32)
    hi = torch.searchsorted(key_s, (tiles + 1) << 32)
    return dict(point_list=point_list, ranges=torch.stack([lo, hi], 1), num_rendered=L)


def blend_tiles(pre, binn, features, bg, W, H, tile_ids=None, chunk=128, group=64):
    """Renders the listed tiles (default all), `group` tiles at a time: every op works on [G, 256 px, n] arrays
    (lists padded to the longest of the group with masked entries), so the host threads get ops worth
    splitting.  -> out (C,H,W), final_T (H,W), n_contrib (H,W) int64, list entries walked"""
    C = features.shape[1]
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    out = torch.zeros(C, gy * TILE, gx * TILE)
    final_T = torch.ones(gy * TILE, gx * TILE)
    n_contrib = torch.zeros(gy * TILE, gx * TILE, dtype=torch.int64)
    pix, conic, opac = pre["pix"], pre["conic"], pre["opacity"]
    plist = binn["point_list"]
    if plist.numel() == 0:
        plist = torch.zeros(1, dtype=torch.int64)
    walked = 0
    yy, xx = torch.meshgrid(torch.arange(TILE, dtype=torch.float32), torch.arange(TILE, dtype=torch.float32),
                            indexing="ij")
    yy, xx = yy.reshape(1, -1, 1), xx.reshape(1, -1, 1)
    tiles = torch.arange(gx * gy) if tile_ids is None else torch.as_tensor(list(tile_ids), dtype=torch.int64)
    ar = torch.arange(chunk)
    for g0 in range(0, tiles.numel(), group):
        tg = tiles[g0:g0 + group]
        G = tg.numel()
        tx, ty = (tg % gx).float().reshape(G, 1, 1), (tg // gx).float().reshape(G, 1, 1)
        px, py = xx + tx * TILE, yy + ty * TILE                                  # (G,256,1)
        inside = ((px < W) & (py < H)).reshape(G, 256)
        r0, r1 = binn["ranges"][tg, 0], binn["ranges"][tg, 1]                     # (G,)
        T = torch.ones(G, 256)
        done = ~inside
        acc = torch.zeros(G, 256, C)
        last = torch.zeros(G, 256, dtype=torch.int64)
        off = 0
        nmax = int((r1 - r0).max()) if G else 0
        while off < nmax and not bool(done.all()):
            pos = r0[:, None] + off + ar[None, :]                                 # (G,n)
            valid = pos < r1[:, None]
            # list entries a tile's block walks: until all of its pixels are done (as the reference's block does)
            walked += int((valid & ~done.all(1)[:, None]).sum())
            ids = plist[torch.where(valid, pos, torch.zeros_like(pos))]          # (G,n)
            dx = pix[ids, 0][:, None, :] - px                                     # (G,256,n)
            dy = pix[ids, 1][:, None, :] - py
            k = conic[ids]                                                        # (G,n,3)
            power = -0.5 * (k[:, None, :, 0] * dx * dx + k[:, None, :, 2] * dy * dy) - k[:, None, :, 1] * dx * dy
            alpha = torch.clamp(opac[ids][:, None, :] * torch.exp(power), max=0.99)
            ok = valid[:, None, :] & (power <= 0) & (alpha >= 1.0 / 255.0) & ~done[:, :, None]
            a_eff = torch.where(ok, alpha, torch.zeros_like(alpha))
            one_m = 1.0 - a_eff
            ones = torch.ones(G, 256, 1)
            T_excl = T[:, :, None] * torch.cumprod(torch.cat([ones, one_m[:, :, :-1]], 2), 2)
            stop = ok & (T_excl * one_m < 1e-4)                                   # the entry that would cross 1e-4 ...
            dead = torch.cumsum(stop.to(torch.int32), 2) > 0                      # ... and everything after it
            a_eff = torch.where(dead, torch.zeros_like(a_eff), a_eff)
            one_m = 1.0 - a_eff
            T_excl = T[:, :, None] * torch.cumprod(torch.cat([ones, one_m[:, :, :-1]], 2), 2)
            wgt = a_eff * T_excl
            acc += torch.bmm(wgt, features[ids])                                  # [G,256,n] x [G,n,C]
            T = T_excl[:, :, -1] * one_m[:, :, -1]
            idx = (ar[None, None, :] + (off + 1)) * (wgt > 0)
            last = torch.maximum(last, idx.amax(2))
            done = done | dead[:, :, -1]
            off += chunk
        res = acc + T[:, :, None] * bg[None, None, :C]                            # (G,256,C)
        for i in range(G):
            ys, xs = int(ty[i]) * TILE, int(tx[i]) * TILE
            out[:, ys:ys + TILE, xs:xs + TILE] = res[i].t().reshape(C, TILE, TILE)
            final_T[ys:ys + TILE, xs:xs + TILE] = T[i].reshape(TILE, TILE)
            n_contrib[ys:ys + TILE, xs:xs + TILE] = last[i].reshape(TILE, TILE)
    return out[:, :H, :W].contiguous(), final_T[:H, :W].contiguous(), n_contrib[:H, :W].contiguous(), walked


def render(scene, cam, W, H, tile_ids=None, timings=None):
    """scene: anything with means3D/scales/rotations/opacities/features/bg (CPU fp32 tensors)."""
    t0 = time.perf_counter()
    pre = preprocess(scene.means3D, scene.scales, scene.rotations, scene.opacities, cam.world_view_transform,
                     cam.full_proj_transform, W, H, cam.tanfovx, cam.tanfovy)
    binn = binning(pre, W, H)
    t1 = time.perf_counter()
    out, final_T, n_contrib, walked = blend_tiles(pre, binn, scene.features, scene.bg, W, H, tile_ids)
    t2 = time.perf_counter()
    if timings is not None:
        timings.update(front_s=t1 - t0, blend_s=t2 - t1, walked=walked)
    return dict(out=out, final_T=final_T, n_contrib=n_contrib, radii=pre["radii"], point_list=binn["point_list"],
                ranges=binn["ranges"], num_rendered=binn["num_rendered"])

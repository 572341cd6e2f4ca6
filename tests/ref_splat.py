"""Independent float64 differentiable Gaussian splat -- TEST INFRASTRUCTURE ONLY.

Purpose: pin the per-Gaussian forward and backward math (CR/cuda_rasterizer/forward.cu:74-255,
backward.cu:20-391) of BOTH the CPU oracle (oracle/sgs_oracle.c) and the HIP kernels against
something that shares no source text with either.  Everything here starts from the RAW inputs
(means3D, scales, rotations, SH coefficients / colours, opacities, camera matrices) and is written
from textbook formulas in torch float64, so torch.autograd supplies the gradients:

  * rotation: Rodrigues form  R = I + 2 r [v]x + 2 [v]x^2  of the quaternion (r, v), used as given
    (the reference does not normalise inside the kernel: forward.cu:127);
  * covariance: Sigma3 = R diag(s)^2 R^T,  Sigma2 = (J W) Sigma3 (J W)^T + 0.3 I  (EWA splatting,
    Zwicker et al. 2001) with J the Jacobian of the pinhole projection at the (clamped) view-space mean;
  * colour: real spherical harmonics from the associated-Legendre recurrence with the
    Condon-Shortley phase (no coefficient table is typed in here; it is checked against the reference's
    own eval_sh fixture in tests/test_ref_splat.py);
  * compositing: dense front-to-back "over" with exclusive cumulative products.

Behaviour of the reference that is NOT textbook and is modelled explicitly (each is a documented quirk,
cited where it is applied): the frustum clamp is a stop-gradient, culling / tile rects / alpha and
transmittance thresholds are masks, the conic backward uses 1/(det^2 + 1e-7).
"""
import math

import torch

F64 = torch.float64


def _skew(v):
    z = torch.zeros_like(v[:, 0])
    return torch.stack([
        torch.stack([z, -v[:, 2], v[:, 1]], -1),
        torch.stack([v[:, 2], z, -v[:, 0]], -1),
        torch.stack([-v[:, 1], v[:, 0], z], -1)], -2)


def quat_to_rot(q):
    """(P,4) quaternion (w,x,y,z), used as given -> (P,3,3)."""
    K = _skew(q[:, 1:])
    eye = torch.eye(3, dtype=q.dtype).expand(q.shape[0], 3, 3)
    return eye + 2.0 * q[:, 0, None, None] * K + 2.0 * K @ K


def real_sh_basis(deg, d):
    """Real spherical harmonics Y_l^m(d), l <= deg, index l*l + l + m; d (P,3) unit vectors."""
    x, y, z = d.unbind(-1)
    cm, sm = [torch.ones_like(x)], [torch.zeros_like(x)]      # Re / Im of (x + i y)^m
    for m in range(1, deg + 1):
        cm.append(cm[m - 1] * x - sm[m - 1] * y)
        sm.append(sm[m - 1] * x + cm[m - 1] * y)
    # Q_l^m = P_l^m(z) / sin(theta)^m, Condon-Shortley phase included
    Q = {}
    for m in range(deg + 1):
        dfact = 1.0
        for k in range(1, 2 * m, 2):
            dfact *= k
        Q[(m, m)] = torch.full_like(z, (-1.0) ** m * dfact)
        if m + 1 <= deg:
            Q[(m + 1, m)] = (2 * m + 1) * z * Q[(m, m)]
        for l in range(m + 2, deg + 1):
            Q[(l, m)] = ((2 * l - 1) * z * Q[(l - 1, m)] - (l + m - 1) * Q[(l - 2, m)]) / (l - m)
    out = []
    for l in range(deg + 1):
        for m in range(-l, l + 1):
            am = abs(m)
            K = math.sqrt((2 * l + 1) / (4 * math.pi) * math.factorial(l - am) / math.factorial(l + am))
            if m == 0:
                out.append(K * Q[(l, 0)])
            elif m > 0:
                out.append(math.sqrt(2.0) * K * Q[(l, am)] * cm[am])
            else:
                out.append(math.sqrt(2.0) * K * Q[(l, am)] * sm[am])
    return torch.stack(out, -1)


def sym3(c6):
    """6-vector xx,xy,xz,yy,yz,zz -> symmetric (P,3,3)."""
    xx, xy, xz, yy, yz, zz = c6.unbind(-1)
    return torch.stack([torch.stack([xx, xy, xz], -1), torch.stack([xy, yy, yz], -1),
                        torch.stack([xz, yz, zz], -1)], -2)


def project(means3D, view, proj, W, H, tanfovx, tanfovy, scales=None, rotations=None,
            scale_modifier=1.0, cov3D_precomp=None, means2D_offset=None):
    """Per-Gaussian geometry.  view / proj are the transposed 4x4 matrices the rasteriser consumes
    (a point transforms as the ROW vector [x y z 1] @ M).  Returns a dict of float64 tensors."""
    P = means3D.shape[0]
    t = means3D @ view[:3, :3] + view[3, :3]                       # view-space mean
    hom = torch.cat([means3D, torch.ones(P, 1, dtype=F64)], 1) @ proj
    inv_w = 1.0 / (hom[:, 3] + 1e-7)                               # forward.cu:200
    ndc = hom[:, :2] * inv_w[:, None]
    if means2D_offset is not None:                                 # the API's dummy "screenspace points"
        ndc = ndc + means2D_offset
    size = torch.tensor([W, H], dtype=F64)
    pix = ((ndc + 1.0) * size - 1.0) * 0.5
    if cov3D_precomp is not None:
        S3 = sym3(cov3D_precomp)
    else:
        R = quat_to_rot(rotations)
        s = scale_modifier * scales
        S3 = R @ torch.diag_embed(s * s) @ R.transpose(1, 2)
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    tz = t[:, 2]
    lim = torch.tensor([1.3 * tanfovx, 1.3 * tanfovy], dtype=F64)
    ratio = t[:, :2] / tz[:, None]
    inside = ratio.abs() <= lim
    # a clamped coordinate is a constant for the backward (gradient multiplier 0, backward.cu:172-173)
    uv = torch.where(inside, t[:, :2], (ratio.clamp(-lim, lim) * tz[:, None]).detach())
    zero = torch.zeros_like(tz)
    J = torch.stack([torch.stack([fx / tz, zero, -fx * uv[:, 0] / (tz * tz)], -1),
                     torch.stack([zero, fy / tz, -fy * uv[:, 1] / (tz * tz)], -1)], -2)
    A = J @ view[:3, :3].transpose(0, 1)                           # (P,2,3): J times the world->view rotation
    S2 = A @ S3 @ A.transpose(1, 2)
    a, b, c = S2[:, 0, 0] + 0.3, S2[:, 0, 1], S2[:, 1, 1] + 0.3
    det = a * c - b * b
    return dict(t=t, depth=tz, pix=pix, a=a, b=b, c=c, det=det, S3=S3, clamp_inside=inside)


class _ConicInverse(torch.autograd.Function):
    """(a,b,c) -> conic (c,-b,a)/det.  Backward = the exact derivative with 1/det^2 replaced by
    1/(det^2 + 1e-7), the reference's guard (backward.cu:200)."""

    @staticmethod
    def forward(ctx, a, b, c):
        det = a * c - b * b
        ctx.save_for_backward(a, b, c)
        return c / det, -b / det, a / det

    @staticmethod
    def backward(ctx, g0, g1, g2):
        a, b, c = ctx.saved_tensors
        det = a * c - b * b
        rho = 1.0 / (det * det + 1e-7)
        # K = adj/det; dK/dtheta = (adj' det - adj det')/det^2, numerator evaluated exactly
        da = rho * (g0 * (-c * c) + g1 * (b * c) + g2 * (det - a * c))
        dc = rho * (g0 * (det - a * c) + g1 * (a * b) + g2 * (-a * a))
        db = rho * (g0 * (2 * b * c) + g1 * (-(det + 2 * b * b)) + g2 * (2 * a * b))
        return da, db, dc


def render(means3D, opacities, view, proj, campos, W, H, tanfovx, tanfovy, bg, scales=None,
           rotations=None, scale_modifier=1.0, cov3D_precomp=None, colors_precomp=None, shs=None,
           sh_degree=0, means2D_offset=None):
    """Differentiable float64 render -> dict(out (C,H,W), radii, depth (P), pix, conic, ...)."""
    g = project(means3D, view, proj, W, H, tanfovx, tanfovy, scales, rotations, scale_modifier,
                cov3D_precomp, means2D_offset)
    P = means3D.shape[0]
    a, b, c, det = g["a"], g["b"], g["c"], g["det"]
    k0, k1, k2 = _ConicInverse.apply(a, b, c)
    # ---- culling, radius, tile rect: masks (no gradient), forward.cu:193-236 / auxiliary.h:46-56
    with torch.no_grad():
        mid = 0.5 * (a + c)
        lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
        radius = torch.ceil(3.0 * torch.sqrt(lam))
        gx, gy = (W + 15) // 16, (H + 15) // 16
        px, py = g["pix"][:, 0], g["pix"][:, 1]
        x0 = torch.clamp(torch.trunc((px - radius) / 16.0), 0, gx)
        x1 = torch.clamp(torch.trunc((px + radius + 15.0) / 16.0), 0, gx)
        y0 = torch.clamp(torch.trunc((py - radius) / 16.0), 0, gy)
        y1 = torch.clamp(torch.trunc((py + radius + 15.0) / 16.0), 0, gy)
        vis = (g["depth"] > 0.2) & (det != 0) & ((x1 - x0) * (y1 - y0) > 0)
        radii = torch.where(vis, radius, torch.zeros_like(radius)).to(torch.int32)
    # ---- colour
    if colors_precomp is not None:
        col = colors_precomp
    else:
        d = means3D - campos
        d = d / d.norm(dim=1, keepdim=True)
        Y = real_sh_basis(sh_degree, d)                            # (P,(deg+1)^2)
        col = torch.einsum("pk,pkc->pc", Y, shs[:, :Y.shape[1], :]) + 0.5
        col = torch.clamp(col, min=0.0)                            # forward.cu:64-69 (mask in the backward)
    C = col.shape[1]
    # ---- dense front-to-back composite over the depth-sorted visible Gaussians
    order = torch.argsort(g["depth"].detach(), stable=True)
    order = order[vis[order]]
    ys, xs = torch.meshgrid(torch.arange(H, dtype=F64), torch.arange(W, dtype=F64), indexing="ij")
    ys, xs = ys.reshape(-1, 1), xs.reshape(-1, 1)                  # (HW,1)
    dx = g["pix"][order, 0][None, :] - xs                          # (HW,K)
    dy = g["pix"][order, 1][None, :] - ys
    power = -0.5 * (k0[order] * dx * dx + k2[order] * dy * dy) - k1[order] * dx * dy
    alpha = opacities.reshape(-1)[order][None, :] * torch.exp(power)
    # min(0.99, .) in the forward (forward.cu:344); the backward has NO mask for it (backward.cu:493-499
    # differentiates o * G whatever the clamp did): straight-through
    alpha = alpha + (torch.clamp(alpha, max=0.99) - alpha).detach()
    with torch.no_grad():
        tx, ty = torch.floor(xs / 16.0), torch.floor(ys / 16.0)
        in_rect = (tx >= x0[order]) & (tx < x1[order]) & (ty >= y0[order]) & (ty < y1[order])
        ok = in_rect & (power <= 0) & (alpha >= 1.0 / 255.0)
    a_eff = torch.where(ok, alpha, torch.zeros_like(alpha))
    one_m = 1.0 - a_eff
    T_excl = torch.cumprod(torch.cat([torch.ones_like(one_m[:, :1]), one_m[:, :-1]], 1), 1)
    with torch.no_grad():                                          # forward.cu:352-357: stop before T < 1e-4
        stop = ok & (T_excl * one_m < 1e-4)
        dead = torch.cumsum(stop.to(torch.int32), 1) > 0
    a_eff = torch.where(dead, torch.zeros_like(a_eff), a_eff)
    one_m = 1.0 - a_eff
    T_excl = torch.cumprod(torch.cat([torch.ones_like(one_m[:, :1]), one_m[:, :-1]], 1), 1)
    wgt = a_eff * T_excl                                           # (HW,K)
    T_final = T_excl[:, -1] * one_m[:, -1] if order.numel() else torch.ones(H * W, dtype=F64)
    out = wgt @ col[order] + T_final[:, None] * bg[None, :C]
    return dict(out=out.t().reshape(C, H, W), radii=radii, depth=g["depth"], pix=g["pix"],
                conic=torch.stack([k0, k1, k2], -1), cov3D=g["S3"], T_final=T_final.reshape(H, W),
                clamp_inside=g["clamp_inside"], vis=vis, colors=col)

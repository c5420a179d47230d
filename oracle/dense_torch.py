"""oracle/dense_torch.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Independent, differentiable, dense O(P*H*W) restatement of the rasterizer in PyTorch (CPU,
float64 by default): no tiles-as-lists, no sorting network -- every pixel looks at every visible
Gaussian in (depth, index) order.  PyTorch autograd of this function is the second opinion for
the hand-derived backward of oracle/tile_ref.c (and therefore of the HIP kernels).

Follows SURVEY.md Appendix A; in-tree anchors: SH basis utils/sh_utils.py:57-112, covariance
utils/general_utils.py:78-110, matrix conventions scene/cameras.py:55-58.
PARITY UNPINNED (see oracle/tile_ref.c header): the reference's rasterizer source is not vendored.

Gradient conventions reproduced from the published algorithm (they are NOT what naive autograd
would do, so they are made explicit here):
  * alpha = min(0.99, op*G): gradient passes straight through the cap
  * the view-space x/y used in the EWA Jacobian are clamped to 1.3*tan(fov/2)*z; when clamped, the
    clamped value is treated as a constant (no gradient to x/y, none to z through the clamp)
  * all selection rules (near cull, power>0, alpha<1/255, T<1e-4 termination, tile rect) are
    piecewise-constant masks
"""
from __future__ import annotations

import math

import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435)


def _sh_rgb(deg, sh, d):
    """sh [P,K,3], d [P,3] unit -> [P,3]"""
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    r = SH_C0 * sh[:, 0]
    if deg > 0:
        r = r - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        r = (r + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * sh[:, 6] +
             SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        r = (r + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10] +
             SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12] +
             SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14] +
             SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return r


def _cov3d(scales, mod, q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    L = R * (mod * scales).unsqueeze(1)
    return L @ L.transpose(1, 2)


def render_dense(*, means3D, opacities, viewmatrix, projmatrix, campos, bg, W, H, tanfovx, tanfovy, sh_degree=0,
                 shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                 scale_modifier=1.0, rect=None, pix_offset=None, dtype=torch.float64):
    """Returns dict(color[3,H,W], depth[1,H,W], alpha[1,H,W], radii[P], pix[P,2]).

    `rect` (optional int tensor [P,4], x0,y0,x1,y1 in tiles; all-zero rows = culled) lets the caller
    impose the integer culling decisions of another implementation so that only the differentiable
    arithmetic is compared.  `pix_offset` ([P,2], requires_grad) is added to the pixel-space means so
    autograd yields dL/d(pixel position) -- the quantity `means2D.grad` reports up to the 0.5*W, 0.5*H
    scaling."""
    c = lambda t: None if t is None else t.to(dtype)  # noqa: E731
    means3D, opacities, shs, colors_precomp = c(means3D), c(opacities).reshape(-1), c(shs), c(colors_precomp)
    scales, rotations, cov3D_precomp = c(scales), c(rotations), c(cov3D_precomp)
    V, PM, campos, bg = c(viewmatrix).reshape(4, 4), c(projmatrix).reshape(4, 4), c(campos).reshape(3), c(bg).reshape(3)
    P = means3D.shape[0]
    ones = torch.ones(P, 1, dtype=dtype)
    ph = torch.cat([means3D, ones], 1)
    pv = ph @ V          # [P,4] view space
    pc = ph @ PM         # clip
    depth = pv[:, 2]
    pw = 1.0 / (pc[:, 3] + 1e-7)
    ndc = pc[:, :2] * pw.unsqueeze(1)
    pix = torch.stack([((ndc[:, 0] + 1) * W - 1) * 0.5, ((ndc[:, 1] + 1) * H - 1) * 0.5], 1)
    if pix_offset is not None:
        pix = pix + pix_offset.to(dtype)

    if cov3D_precomp is not None:
        s6 = cov3D_precomp
        S = torch.stack([s6[:, 0], s6[:, 1], s6[:, 2], s6[:, 1], s6[:, 3], s6[:, 4], s6[:, 2], s6[:, 4], s6[:, 5]],
                        1).reshape(-1, 3, 3)
    else:
        S = _cov3d(scales, scale_modifier, rotations)

    fx, fy = W / (2 * tanfovx), H / (2 * tanfovy)
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    tz = pv[:, 2]
    txtz, tytz = pv[:, 0] / tz, pv[:, 1] / tz
    cx, cy = (txtz < -limx) | (txtz > limx), (tytz < -limy) | (tytz > limy)
    tx = torch.where(cx, (txtz.clamp(-limx, limx) * tz).detach(), pv[:, 0])
    ty = torch.where(cy, (tytz.clamp(-limy, limy) * tz).detach(), pv[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz), zero, fy / tz, -(fy * ty) / (tz * tz)], 1).reshape(-1, 2, 3)
    Wr = V[:3, :3].t()   # conventional W2C rotation
    T = J @ Wr
    cov = T @ S @ T.transpose(1, 2)
    a, b, cc = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * cc - b * b
    conic = torch.stack([cc / det, -b / det, a / det], 1)
    mid = 0.5 * (a + cc)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3 * torch.sqrt(lam)).detach()

    gx, gy = (W + 15) // 16, (H + 15) // 16
    if rect is None:
        def cl(v, hi):
            return torch.clamp(v, 0, hi).to(torch.int64)
        p = pix.detach()
        rect = torch.stack([cl((p[:, 0] - radius) / 16, gx), cl((p[:, 1] - radius) / 16, gy),
                            cl((p[:, 0] + radius + 15) / 16, gx), cl((p[:, 1] + radius + 15) / 16, gy)], 1)
        rect[(depth.detach() <= 0.2) | (det.detach() == 0)] = 0
    rect = rect.to(torch.int64)
    visible = ((rect[:, 2] - rect[:, 0]) * (rect[:, 3] - rect[:, 1])) > 0

    if colors_precomp is not None:
        rgb = colors_precomp
    else:
        d = means3D - campos
        d = d / d.norm(dim=1, keepdim=True)
        rgb = torch.clamp_min(_sh_rgb(sh_degree, shs, d) + 0.5, 0.0)

    # (depth, index) order of the visible set.  The depth key is the FLOAT32 view depth, as in the
    # tile sort (ties in float32 must stay ties here).
    idx = torch.nonzero(visible).reshape(-1)
    d32 = depth.detach()[idx].to(torch.float32)
    order = idx[torch.sort(d32, stable=True).indices]
    n = order.numel()
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dtype), torch.arange(W, dtype=dtype), indexing="ij")
    if n == 0:
        color = bg.reshape(3, 1, 1).expand(3, H, W).clone()
        zero_img = torch.zeros(1, H, W, dtype=dtype)
        return dict(color=color, depth=zero_img, alpha=zero_img.clone(), radii=(radius * visible).to(torch.int32),
                    pix=pix)
    po = pix[order]
    dx = po[:, 0].reshape(n, 1, 1) - xs
    dy = po[:, 1].reshape(n, 1, 1) - ys
    co = conic[order]
    power = -0.5 * (co[:, 0].reshape(n, 1, 1) * dx * dx + co[:, 2].reshape(n, 1, 1) * dy * dy) \
        - co[:, 1].reshape(n, 1, 1) * dx * dy
    G = torch.exp(torch.clamp(power, max=0.0))
    a_raw = opacities[order].reshape(n, 1, 1) * G
    alpha = a_raw + (torch.clamp(a_raw, max=0.99) - a_raw).detach()
    # membership of the pixel's tile in the Gaussian's rect
    tx_ = (xs // 16).to(torch.int64)
    ty_ = (ys // 16).to(torch.int64)
    ro = rect[order]
    in_rect = ((tx_ >= ro[:, 0].reshape(n, 1, 1)) & (tx_ < ro[:, 2].reshape(n, 1, 1)) &
               (ty_ >= ro[:, 1].reshape(n, 1, 1)) & (ty_ < ro[:, 3].reshape(n, 1, 1)))
    keep = in_rect & (power.detach() <= 0) & (alpha.detach() >= 1.0 / 255.0)
    alpha = torch.where(keep, alpha, torch.zeros_like(alpha))
    # termination: first i with T_i (1 - alpha_i) < 1e-4 stops the pixel (that one is not blended)
    one_m = 1 - alpha
    Tincl = torch.cumprod(one_m, dim=0)
    stop = (Tincl.detach() < 1e-4) & keep
    stopped = torch.cumsum(stop.to(torch.int32), dim=0) > 0
    alpha = torch.where(stopped, torch.zeros_like(alpha), alpha)
    one_m = 1 - alpha
    Tincl = torch.cumprod(one_m, dim=0)
    Texcl = torch.cat([torch.ones(1, H, W, dtype=dtype), Tincl[:-1]], 0)
    wgt = alpha * Texcl
    color = (wgt.unsqueeze(1) * rgb[order].reshape(n, 3, 1, 1)).sum(0) + Tincl[-1].unsqueeze(0) * bg.reshape(3, 1, 1)
    dep = (wgt * depth[order].reshape(n, 1, 1)).sum(0, keepdim=True)
    alp = wgt.sum(0, keepdim=True)
    return dict(color=color, depth=dep, alpha=alp, radii=(radius * visible).to(torch.int32), pix=pix,
                final_T=Tincl[-1])

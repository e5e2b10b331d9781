"""ORACLE (test infrastructure, CPU, fp32) — forward Gaussian-splatting rasteriser.  **PARITY UNPINNED.**

The reference renders through ``diff_gaussian_rasterization`` (``core/gs.py:7-10,57-83``), a third-party CUDA extension
that ``install.sh:3`` clones at an unpinned HEAD of ``ashawkey/diff-gaussian-rasterization`` and that is absent from
``/root/reference`` — it can neither be imported nor built here, and the reference holds no test or golden for it.  This
file therefore restates the PUBLISHED forward algorithm of 3-D Gaussian Splatting (Kerbl et al. 2023, forward pass of
the tile rasteriser; the fork adds accumulated depth and alpha outputs) for the reference's call site:
``sh_degree=0``, ``colors_precomp`` given, ``scale_modifier=1``, square image, ``tanfovx = tanfovy``, row-vector
``viewmatrix`` / ``projmatrix`` (``p_row @ M``), background colour blended with the final transmittance.

Per Gaussian (mean m, scales s, quaternion q = (r, x, y, z) used AS GIVEN — not re-normalised, opacity o, colour c):
  p_view = [m,1] @ V ;  culled if p_view.z <= 0.2 ;  p_hom = [m,1] @ VP ;  p_ndc = p_hom.xyz / (p_hom.w + 1e-7)
  Sigma3 = R diag(s^2) R^T  (R from q) ;  t = p_view with t.xy/t.z clamped to +-1.3 tan(fov/2)
  J = [[f/tz, 0, -f tx/tz^2], [0, f/tz, -f ty/tz^2]] ,  f = W / (2 tan) ;  Sigma2 = J W3 Sigma3 W3^T J^T + 0.3 I  (W3 = V[:3,:3]^T)
  det = a c - b^2 (skip if 0) ; conic = (c, -b, a)/det ; radius = ceil(3 sqrt(max eigenvalue)) with mid^2 - det floored at 0.1
  centre px = ((ndc + 1) W - 1) / 2 ; tile rect = [floor((px - r)/16), floor((px + r + 15)/16)) clamped to the 16x16-tile grid
Per pixel (integer coordinates, no half-pixel offset), over the Gaussians whose tile rect contains the pixel's tile,
sorted by (p_view.z, index): power = -0.5 (A dx^2 + C dy^2) - B dx dy ; skip if power > 0 ; alpha = min(0.99, o e^power) ;
skip if alpha < 1/255 ; stop BEFORE blending if T (1 - alpha) < 1e-4 ; colour += c alpha T ; depth += z alpha T ;
weight += alpha T ; T *= 1 - alpha.  Output colour + T * background, accumulated alpha weight, depth.
Validated on analytic cases in tests/test_gs_cpu.py (single isotropic Gaussian against the closed form, two overlapping
Gaussians and their order swap, culling, tile truncation).
"""
import math

import torch

TILE = 16


def quat_to_rot(q):
    """[N,4] (r,x,y,z), not normalised -> [N,3,3] (the standard formula applied to the raw components)."""
    r, x, y, z = q.unbind(-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.reshape(-1, 3, 3)


def preprocess(means, scales, rots, view, view_proj, size, tan_half_fov):
    """-> dict(valid, depth, xy [N,2], conic [N,3], radius, rect (x0,y0,x1,y1) in tiles)."""
    N = means.shape[0]
    ones = torch.ones(N, 1, dtype=means.dtype)
    ph = torch.cat([means, ones], dim=1)
    p_view = (ph @ view)[:, :3]
    p_hom = ph @ view_proj
    p_ndc = p_hom[:, :3] / (p_hom[:, 3:4] + 1e-7)
    valid = p_view[:, 2] > 0.2
    R = quat_to_rot(rots)
    S2 = torch.diag_embed(scales * scales)
    Sigma3 = R @ S2 @ R.transpose(1, 2)
    focal = size / (2.0 * tan_half_fov)
    lim = 1.3 * tan_half_fov
    tz = p_view[:, 2]
    tx = torch.clamp(p_view[:, 0] / tz, -lim, lim) * tz
    ty = torch.clamp(p_view[:, 1] / tz, -lim, lim) * tz
    J = torch.zeros(N, 2, 3, dtype=means.dtype)
    J[:, 0, 0] = focal / tz
    J[:, 0, 2] = -focal * tx / (tz * tz)
    J[:, 1, 1] = focal / tz
    J[:, 1, 2] = -focal * ty / (tz * tz)
    W3 = view[:3, :3].transpose(0, 1)                      # world -> view rotation for column vectors
    T = J @ W3
    cov = T @ Sigma3 @ T.transpose(1, 2)
    a = cov[:, 0, 0] + 0.3
    b = cov[:, 0, 1]
    c = cov[:, 1, 1] + 0.3
    det = a * c - b * b
    valid = valid & (det != 0)
    det_safe = torch.where(det == 0, torch.ones_like(det), det)
    conic = torch.stack([c / det_safe, -b / det_safe, a / det_safe], dim=1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam))
    xy = ((p_ndc[:, :2] + 1.0) * size - 1.0) * 0.5
    grid = (size + TILE - 1) // TILE
    x0 = torch.clamp(torch.floor((xy[:, 0] - radius) / TILE), 0, grid)
    y0 = torch.clamp(torch.floor((xy[:, 1] - radius) / TILE), 0, grid)
    x1 = torch.clamp(torch.floor((xy[:, 0] + radius + TILE - 1) / TILE), 0, grid)
    y1 = torch.clamp(torch.floor((xy[:, 1] + radius + TILE - 1) / TILE), 0, grid)
    valid = valid & ((x1 - x0) * (y1 - y0) > 0)
    rect = torch.stack([x0, y0, x1, y1], dim=1).long()
    return dict(valid=valid, depth=tz, xy=xy, conic=conic, radius=radius, rect=rect)


@torch.no_grad()
def render(means, opacity, scales, rots, colors, view, view_proj, size, tan_half_fov, bg):
    """One view.  means [N,3], opacity [N,1], scales [N,3], rots [N,4], colors [N,3]; view / view_proj [4,4] (row-vector
    convention); bg [3].  -> (image [3,size,size] (NOT clamped: core/gs.py:84 clamps afterwards), alpha [1,size,size],
    depth [1,size,size])."""
    pp = preprocess(means.float(), scales.float(), rots.float(), view.float(), view_proj.float(), size, tan_half_fov)
    idx = torch.nonzero(pp["valid"]).flatten()
    order = idx[torch.argsort(pp["depth"][idx], stable=True)]          # front to back; ties keep index order
    C = torch.zeros(3, size, size)
    D = torch.zeros(size, size)
    Wt = torch.zeros(size, size)
    T = torch.ones(size, size)
    done = torch.zeros(size, size, dtype=torch.bool)
    ys, xs = torch.meshgrid(torch.arange(size, dtype=torch.float32), torch.arange(size, dtype=torch.float32), indexing="ij")
    for g in order.tolist():
        x0, y0, x1, y1 = (int(v) for v in pp["rect"][g])
        sl = (slice(y0 * TILE, min(y1 * TILE, size)), slice(x0 * TILE, min(x1 * TILE, size)))
        dx = pp["xy"][g, 0] - xs[sl]
        dy = pp["xy"][g, 1] - ys[sl]
        A, B, Cc = pp["conic"][g]
        power = -0.5 * (A * dx * dx + Cc * dy * dy) - B * dx * dy
        alpha = torch.clamp(opacity[g, 0] * torch.exp(power), max=0.99)
        hit = (power <= 0) & (alpha >= 1.0 / 255.0) & ~done[sl]
        testT = T[sl] * (1 - alpha)
        stop = hit & (testT < 1e-4)
        done[sl] |= stop
        hit = hit & ~stop
        w = torch.where(hit, alpha * T[sl], torch.zeros_like(alpha))
        C[:, sl[0], sl[1]] += colors[g].float().view(3, 1, 1) * w
        D[sl] += pp["depth"][g] * w
        Wt[sl] += w
        T[sl] = torch.where(hit, testT, T[sl])
    img = C + T.unsqueeze(0) * bg.float().view(3, 1, 1)
    return img, Wt.unsqueeze(0), D.unsqueeze(0)


@torch.no_grad()
def render_views(gaussians, cam_view, cam_view_proj, size, fovy_deg, bg):
    """gaussians [N,14] (pos 3, opacity 1, scale 3, rotation 4, rgb 3); cam_* [V,4,4] -> images [V,3,S,S] clamped to
    [0,1] (core/gs.py:84), alphas [V,1,S,S]."""
    tan = math.tan(0.5 * math.radians(fovy_deg))
    imgs, alphas = [], []
    for v in range(cam_view.shape[0]):
        im, al, _ = render(gaussians[:, 0:3], gaussians[:, 3:4], gaussians[:, 4:7], gaussians[:, 7:11], gaussians[:, 11:14],
                           cam_view[v], cam_view_proj[v], size, tan, bg)
        imgs.append(im.clamp(0, 1))
        alphas.append(al)
    return torch.stack(imgs), torch.stack(alphas)

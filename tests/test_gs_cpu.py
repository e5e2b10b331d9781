"""Analytic checks of the splatting oracle (oracle/gs_ref.py — PARITY UNPINNED: the reference's rasteriser is an absent,
unpinned third-party extension, so the restatement of the published algorithm is validated against closed forms)."""
import math

import torch

from oracle.gs_ref import render, preprocess, quat_to_rot

TAN = math.tan(0.5 * math.radians(39.6))


def _proj(znear=0.5, zfar=2.5):
    P = torch.zeros(4, 4)
    P[0, 0] = P[1, 1] = 1 / TAN
    P[2, 2] = (zfar + znear) / (zfar - znear)
    P[3, 2] = -(zfar * znear) / (zfar - znear)
    P[2, 3] = 1
    return P


def _g(pos, s, o, rgb, q=(1.0, 0.0, 0.0, 0.0)):
    return (torch.tensor([pos], dtype=torch.float32), torch.tensor([[o]]), torch.tensor([[s, s, s]]), torch.tensor([q]),
            torch.tensor([rgb]))


def test_single_isotropic_gaussian_closed_form():
    S, d, s, o = 64, 1.5, 0.05, 0.8
    view, vp = torch.eye(4), _proj()
    m, op, sc, q, rgb = _g((0.0, 0.0, d), s, o, (0.9, 0.2, 0.1))
    bg = torch.tensor([0.5, 0.5, 0.5])
    img, alpha, depth = render(m, op, sc, q, rgb, view, vp, S, TAN, bg)
    f = S / (2 * TAN)
    var = (f * s / d) ** 2 + 0.3
    c = (S - 1) / 2.0
    ys, xs = torch.meshgrid(torch.arange(S, dtype=torch.float32), torch.arange(S, dtype=torch.float32), indexing="ij")
    a = torch.clamp(o * torch.exp(-0.5 * ((xs - c) ** 2 + (ys - c) ** 2) / var), max=0.99)
    a = torch.where(a >= 1 / 255, a, torch.zeros_like(a))
    radius = math.ceil(3 * math.sqrt(var))                  # both eigenvalues equal var (mid^2 - det = 0 -> floored at 0.1)
    radius = math.ceil(3 * math.sqrt(var + math.sqrt(0.1)))
    x0, x1 = max(0, int((c - radius) // 16)), min(S // 16, int((c + radius + 15) // 16))
    mask = torch.zeros(S, S, dtype=torch.bool)
    mask[x0 * 16:x1 * 16, x0 * 16:x1 * 16] = True
    a = torch.where(mask, a, torch.zeros_like(a))
    exp = torch.tensor([0.9, 0.2, 0.1]).view(3, 1, 1) * a + (1 - a) * bg.view(3, 1, 1)
    assert torch.allclose(img, exp, atol=1e-5)
    assert torch.allclose(alpha[0], a, atol=1e-6) and torch.allclose(depth[0], d * a, atol=1e-5)


def test_two_gaussians_composite_front_to_back_and_order_swap():
    S = 32
    view, vp, bg = torch.eye(4), _proj(), torch.zeros(3)
    m = torch.tensor([[0.0, 0.0, 1.0], [0.0, 0.0, 1.4]])
    op = torch.tensor([[0.6], [0.9]])
    sc = torch.full((2, 3), 0.2)
    q = torch.tensor([[1.0, 0, 0, 0]] * 2)
    rgb = torch.tensor([[1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    img, alpha, _ = render(m, op, sc, q, rgb, view, vp, S, TAN, bg)
    c = S // 2
    pp = preprocess(m, sc, q, view, vp, S, TAN)

    def a_at(i, x, y):
        dx, dy = pp["xy"][i, 0] - x, pp["xy"][i, 1] - y
        A, B, Cc = pp["conic"][i]
        return min(0.99, float(op[i, 0] * torch.exp(-0.5 * (A * dx * dx + Cc * dy * dy) - B * dx * dy)))
    a1, a2 = a_at(0, c, c), a_at(1, c, c)
    assert abs(float(img[0, c, c]) - a1) < 1e-5 and abs(float(img[2, c, c]) - a2 * (1 - a1)) < 1e-5
    assert abs(float(alpha[0, c, c]) - (a1 + a2 * (1 - a1))) < 1e-5
    m2 = m.clone()
    m2[0, 2], m2[1, 2] = 1.4, 1.0                       # swap depths: blue is now in front
    img2, _, _ = render(m2, op, sc, q, rgb, view, vp, S, TAN, bg)
    pp2 = preprocess(m2, sc, q, view, vp, S, TAN)
    dxb = pp2["xy"][1] - torch.tensor([float(c), float(c)])
    ab = min(0.99, float(op[1, 0] * torch.exp(-0.5 * (pp2["conic"][1, 0] * dxb[0] ** 2 + pp2["conic"][1, 2] * dxb[1] ** 2)
                                              - pp2["conic"][1, 1] * dxb[0] * dxb[1])))
    assert abs(float(img2[2, c, c]) - ab) < 1e-5 and float(img2[0, c, c]) < float(img[0, c, c])


def test_near_cull_saturation_stop_and_rotation_formula():
    S = 32
    view, vp, bg = torch.eye(4), _proj(), torch.ones(3)
    m, op, sc, q, rgb = _g((0.0, 0.0, 0.15), 0.1, 0.9, (0.0, 0.0, 0.0))      # behind the 0.2 near cut: invisible
    img, alpha, _ = render(m, op, sc, q, rgb, view, vp, S, TAN, bg)
    assert torch.equal(img, torch.ones(3, S, S)) and float(alpha.abs().max()) == 0.0
    # ten layers of opacity 0.8: blending stops BEFORE the layer that would drop the transmittance under 1e-4
    n = 10
    m = torch.stack([torch.tensor([0.0, 0.0, 1.0 + 0.01 * i]) for i in range(n)])
    sc, q = torch.full((n, 3), 0.5), torch.tensor([[1.0, 0, 0, 0]] * n)
    img, alpha, _ = render(m, torch.full((n, 1), 0.8), sc, q, torch.zeros(n, 3), view, vp, S, TAN, bg)
    pp = preprocess(m, sc, q, view, vp, S, TAN)
    c = S // 2
    T, w = 1.0, 0.0
    for i in range(n):
        dx, dy = float(pp["xy"][i, 0]) - c, float(pp["xy"][i, 1]) - c
        A, B, Cc = (float(v) for v in pp["conic"][i])
        a = min(0.99, 0.8 * math.exp(-0.5 * (A * dx * dx + Cc * dy * dy) - B * dx * dy))
        if T * (1 - a) < 1e-4:
            break
        w += a * T
        T *= 1 - a
    assert i == 5 and abs(float(alpha[0, c, c]) - w) < 1e-6 and abs(float(img[0, c, c]) - T) < 1e-6    # black layers on white
    # a 90-degree rotation about z swaps the x and y extents of an anisotropic Gaussian
    R = quat_to_rot(torch.tensor([[math.sqrt(0.5), 0.0, 0.0, math.sqrt(0.5)]]))[0]
    assert torch.allclose(R @ torch.tensor([1.0, 0.0, 0.0]), torch.tensor([0.0, 1.0, 0.0]), atol=1e-6)
    scs = torch.tensor([[0.3, 0.05, 0.05]])
    p0 = preprocess(torch.tensor([[0.0, 0.0, 1.5]]), scs, torch.tensor([[1.0, 0, 0, 0]]), view, vp, 64, TAN)
    p1 = preprocess(torch.tensor([[0.0, 0.0, 1.5]]), scs, torch.tensor([[math.sqrt(0.5), 0.0, 0.0, math.sqrt(0.5)]]), view, vp, 64, TAN)
    assert torch.allclose(p0["conic"][0, 0], p1["conic"][0, 2], rtol=1e-4) and torch.allclose(p0["conic"][0, 2], p1["conic"][0, 0], rtol=1e-4)


def test_oracle_closed_forms_rotated_anisotropic_and_tile_boundary_order():
    """The two closed forms the GPU tier holds the HIP rasteriser to (tests/test_gs_gpu.py) hold for the oracle restatement too:
    a rotated anisotropic Gaussian on the optical axis, and depth-ordered compositing of two off-axis Gaussians whose centres sit on
    a 16-pixel tile boundary, in both memory orders."""
    import math
    import torch
    from oracle.gs_ref import render_views
    from videomv_amd.gs import GaussianRenderer
    from tests.test_gs_gpu import _closed_form_cov2d, _alpha
    S, d, o, th = 64, 1.5, 0.9, math.radians(35.0)
    sx, sy, sz = 0.09, 0.03, 0.05
    r = GaussianRenderer(output_size=S)
    f = S / (2 * r.tan_half_fov)
    c, s = math.cos(th), math.sin(th)
    S3 = [[c * c * sx * sx + s * s * sy * sy, c * s * (sx * sx - sy * sy), 0.0],
          [c * s * (sx * sx - sy * sy), s * s * sx * sx + c * c * sy * sy, 0.0], [0.0, 0.0, sz * sz]]
    cov = _closed_form_cov2d(0.0, 0.0, d, S3, f)
    one = torch.tensor([[0.0, 0.0, d, o, sx, sy, sz, math.cos(th / 2), 0.0, 0.0, math.sin(th / 2), 0.2, 0.7, 0.4]])
    _, al = render_views(one, torch.eye(4).view(1, 4, 4), r.proj_matrix.view(1, 4, 4), S, 39.6, torch.zeros(3))
    cx = cy = (S - 1) / 2.0
    assert max(abs(float(al[0, 0, py, px]) - _alpha(px, py, cx, cy, cov, o)) for py in range(24, 41) for px in range(24, 41)) < 2e-6
    tan = r.tan_half_fov
    ndc_x, ndc_y = (2 * 15.5 + 1) / S - 1.0, (2 * 31.5 + 1) / S - 1.0
    rows, info = [], []
    for z, sg, og, col in ((1.2, 0.06, 0.7, (1.0, 0.1, 0.2)), (1.9, 0.10, 0.8, (0.1, 0.9, 0.3))):
        x, y = ndc_x * tan * z, ndc_y * tan * z
        rows.append([x, y, z, og, sg, sg, sg, 1.0, 0.0, 0.0, 0.0, *col])
        info.append((_closed_form_cov2d(x, y, z, [[sg * sg, 0, 0], [0, sg * sg, 0], [0, 0, sg * sg]], f), og, col))
    for order in (rows, rows[::-1]):
        img, _ = render_views(torch.tensor(order), torch.eye(4).view(1, 4, 4), r.proj_matrix.view(1, 4, 4), S, 39.6, torch.zeros(3))
        for py in range(28, 36):
            for px in range(12, 20):
                a0, a1 = _alpha(px, py, 15.5, 31.5, info[0][0], info[0][1]), _alpha(px, py, 15.5, 31.5, info[1][0], info[1][1])
                for ch in range(3):
                    assert abs(float(img[0, ch, py, px]) - (info[0][2][ch] * a0 + info[1][2][ch] * a1 * (1 - a0))) < 2e-6

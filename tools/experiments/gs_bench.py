#!/usr/bin/env python
"""Rasteriser alone at the LGM-refined step's size (GPU only): 65 536 Gaussians x 24 views at 512 x 512, synthetic Gaussians sized so
that a view has ~0.88 M (tile, Gaussian) instances as bench.py's LGM leg reports.   python tools/experiments/gs_bench.py [reps]
Under `rocprofv3 --kernel-trace --stats` the per-kernel split of the batched pass."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videomv_amd.gs import GaussianRenderer


def scene(n, seed, smax):
    g = torch.Generator().manual_seed(seed)
    pos = (torch.rand(n, 3, generator=g) - 0.5) * 1.0
    opacity = torch.rand(n, 1, generator=g)
    scale = 0.003 + smax * torch.rand(n, 3, generator=g)
    rot = torch.randn(n, 4, generator=g)
    rot = rot / rot.norm(dim=1, keepdim=True)
    rgb = torch.rand(n, 3, generator=g)
    return torch.cat([pos, opacity, scale, rot, rgb], dim=1)


def cams(views, P, dist=1.5, elevation=15.0):
    cv, cvp = [], []
    for i in range(views):
        az, el = math.radians(360.0 * i / views + 10.0), math.radians(elevation)
        pos = dist * torch.tensor([math.cos(el) * math.sin(az), math.sin(el), math.cos(el) * math.cos(az)])
        fwd = -pos / pos.norm()
        right = torch.linalg.cross(fwd, torch.tensor([0.0, 1.0, 0.0])); right = right / right.norm()
        down = torch.linalg.cross(fwd, right)
        c2w = torch.eye(4)
        c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, down, fwd, pos
        view = torch.inverse(c2w).transpose(0, 1)
        cv.append(view); cvp.append(view @ P)
    return torch.stack(cv), torch.stack(cvp)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    r = GaussianRenderer(output_size=512, fovy=49.1)
    g = scene(65536, 3, float(os.environ.get("GS_SMAX", "0.03"))).cuda().unsqueeze(0)
    cv, cvp = cams(24, r.proj_matrix)
    cv, cvp = cv.unsqueeze(0).cuda(), cvp.unsqueeze(0).cuda()
    for mode in ("1", "0"):
        os.environ["VMV_GS_BATCH"] = mode
        out = r.render(g, cv, cvp, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = r.render(g, cv, cvp, None)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"VMV_GS_BATCH={mode}: {1e3 * dt:.3f} ms per 24 views, instances {sum(r.last_num_rendered)} ({sum(r.last_num_rendered) // 24} per view), "
              f"image sum {float(out['image'].double().sum()):.6f} alpha sum {float(out['alpha'].double().sum()):.6f}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""proj_in at the two large levels: plain gemm_rs on a normalised tensor against gemm_rs with the GroupNorm folded in (gn_table);
and the passes it replaces.  us per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videomv_amd import _lib as L, ops

BF = L.elem()


def bench(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1000


dev = "cuda"
for C, HW in ((320, 2560), (640, 640)):
    M = 48 * HW
    x = torch.randn(M, C, device=dev).to(BF)
    y = torch.empty_like(x)
    o = torch.empty(M, C, device=dev, dtype=BF)
    w = (torch.randn(C, C, device=dev) * C ** -0.5).to(BF)
    b = torch.randn(C, device=dev)
    g, be = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    part = torch.zeros(ops.gn_partial_floats(M, HW, C) + 64, device=dev)
    tab = torch.zeros(48, 2, C, device=dev)
    S = ops.Stream(record=False)
    gp = ops.gn_params(x, C, C, M, HW, part, g, be, 1e-6, False, y, C)
    gt = ops.gn_params(x, C, C, M, HW, part, g, be, 1e-6, False, tab, C)
    S.groupnorm_stats(gp); S.groupnorm_table(gt)
    p0 = ops.gemm_params(M, C, ops.linear_segs([(y, C, C)]), w, o, C, bias=b)
    p1 = ops.gemm_params(M, C, ops.linear_segs([(x, C, C)]), w, o, C, bias=b, gn_table=tab, gn_rows_per_stat=HW)
    t_apply, t_table = bench(lambda: S.groupnorm_apply(gp)), bench(lambda: S.groupnorm_table(gt))
    t0, t1 = bench(lambda: S.gemm(p0)), bench(lambda: S.gemm(p1))
    print(f"C={C} M={M}: apply {t_apply:6.1f}  table {t_table:5.1f}  proj_in plain {t0:6.1f}  proj_in folded {t1:6.1f} us   "
          f"(apply + plain {t_apply + t0:6.1f} vs table + folded {t_table + t1:6.1f})", flush=True)

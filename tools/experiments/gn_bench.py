#!/usr/bin/env python
"""GroupNorm statistics pass at the UNet's L0 shapes: all-frame (2 stat groups of 61 440 rows) against per-frame (48 x 2560), the
int64 totals against per-chunk partials, over the chunk size.  us per launch."""
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
    rows = 2 * 24 * HW
    x = torch.randn(rows, C, device=dev).to(BF)
    y = torch.empty_like(x)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    S = ops.Stream(record=False)
    for rps in (24 * HW, HW):
        for cr in (None, 120, 60, 30):
            cr_ = cr or ops.gn_chunk_rows(rps, C)
            if cr_ > rps: continue
            part = torch.zeros(ops.gn_partial_floats(rows, rps, C, cr_) + 64, device=dev)
            tot = torch.zeros(64 * ops.GN_TOT, device=dev, dtype=torch.int64)
            pp = ops.gn_params(x, C, C, rows, rps, part, g, b, 1e-5, True, y, C, chunk_rows=cr_)
            pt = ops.gn_params(x, C, C, rows, rps, part, g, b, 1e-5, True, y, C, chunk_rows=cr_, totals=tot)
            tp = bench(lambda: S.groupnorm_stats(pp))
            tt = bench(lambda: S.groupnorm_stats(pt)) if (rps + cr_ - 1) // cr_ <= 4096 else float("nan")
            ta = bench(lambda: S.groupnorm_apply(pp))
            print(f"C={C} rows/stat={rps:6d} chunk_rows={cr_:4d} nchunk={(rps + cr_ - 1) // cr_:5d}: stats(partials) {tp:6.1f}  stats(totals) {tt:6.1f}  apply(partials) {ta:6.1f} us", flush=True)

print("one-launch form (vmv_groupnorm_fused): us, GB/s of read + write")
for name, (rps, C, nstat) in {"L1 frame 640": (640, 640, 48), "L1 frame concat 1280": (640, 1280, 48), "L2 frame 1280": (160, 1280, 48),
                              "L2 concat 2560": (160, 2560, 48), "L3 frame 1280": (40, 1280, 48), "L3 all-frame 1280": (960, 1280, 2)}.items():
    rows = rps * nstat
    x = torch.randn(rows, C, device=dev).to(BF)
    y = torch.empty_like(x)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    part = torch.zeros(ops.gn_partial_floats(rows, rps, C) + 64, device=dev)
    S = ops.Stream(record=False)
    cols = ops.gn_fused_cols(rps, C)
    pf = ops.gn_params(x, C, C, rows, rps, part, g, b, 1e-5, True, y, C)
    tf = bench(lambda: S.groupnorm_fused(pf, cols))
    p2 = ops.gn_params(x, C, C, rows, rps, part, g, b, 1e-5, True, y, C)
    t2 = bench(lambda: (S.groupnorm_stats(p2), S.groupnorm_apply(p2)))
    print(f"{name:22s} cols={cols:3d} fused {tf:6.1f} us ({2 * rows * C * 2 / tf / 1e3:6.0f} GB/s)   stats+apply {t2:6.1f} us", flush=True)

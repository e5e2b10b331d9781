#!/usr/bin/env python
"""GPU: where the one-launch GroupNorm (vmv_groupnorm_fused) spends its time — s_memtime stamps of block (0, 0) at the phase edges
(VMV_GNF_STAMP=1) next to the launch's event time, on the plan's small-level shapes."""
import os, sys
os.environ["VMV_GNF_STAMP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videomv_amd import _lib as L, ops

BF = L.elem()
dev = "cuda"


def bench(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1000


print("phases: load+stage+colsum | mean totals | sq-dev pass | rstd totals | scale/shift | apply+store   (s_memtime ticks of block 0; event us per launch)")
for name, (rps, C, nstat) in {"L3 all-frame 1280 (tconv)": (960, 1280, 2), "L3 frame 1280": (40, 1280, 48), "L3 frame concat 2560": (40, 2560, 48), "L2 frame 1280": (160, 1280, 48),
                              "L2 concat 2560": (160, 2560, 48), "L1 frame 640": (640, 640, 48), "32x32 L2 all-frame 1280": (1536, 1280, 2), "32x32 L3 all-frame": (384, 1280, 2)}.items():
    rows = rps * nstat
    x = torch.randn(rows, C, device=dev).to(BF)
    y = torch.empty_like(x)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    part = torch.zeros(4096, device=dev)
    S = ops.Stream(record=False)
    cols = ops.gn_fused_cols(rps, C)
    if not cols:
        print(f"{name:28s} does not fit on chip"); continue
    pf = ops.gn_params(x, C, C, rows, rps, part, g, b, 1e-5, True, y, C)
    t = bench(lambda: S.groupnorm_fused(pf, cols))
    st = part.view(torch.int64)[:7].cpu().tolist()
    d = [st[i + 1] - st[i] for i in range(6)]
    print(f"{name:28s} cols={cols:3d} grid=({C // cols},{nstat}) {t:6.1f} us  ticks {d}  total {st[6] - st[0]}", flush=True)

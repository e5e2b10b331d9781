import torch, sys, os
sys.path.insert(0, os.getcwd())
from videomv_amd import _lib as L, ops
S = ops.Stream(record=False)
def bench(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1000
for rows, rps, C in ((122880, 2560, 320), (122880, 61440, 320), (30720, 640, 640), (30720, 15360, 640)):
    x = torch.randn(rows, C, device="cuda").to(L.elem())
    y = torch.empty_like(x)
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    part = torch.zeros(ops.gn_partial_floats(rows, rps, C) + 64, device="cuda")
    tot = torch.zeros(64 * 64, device="cuda")
    allf = rps > 10000
    p = ops.gn_params(x, C, C, rows, rps, part, g, b, 1e-5, True, y, C, totals=tot if allf else None)
    ts = bench(lambda: S.groupnorm_stats(p)); ta = bench(lambda: S.groupnorm_apply(p))
    mb = rows * C * 2 / 1e6
    print(f"rows {rows} rps {rps} C {C}: stats {ts:.1f} us ({mb/ts/1e0:.2f} GB/ms = {mb/ts:.2f} TB/s) apply {ta:.1f} us ({2*mb/ta:.2f} TB/s)")

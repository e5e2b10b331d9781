"""Debug helper: error map of one GEMM tile configuration vs torch (GPU only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videomv_amd import _lib as L, ops
M, N, K, tile = (int(v) for v in sys.argv[1:5])
torch.manual_seed(0)
x = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
b = torch.randn(N, device="cuda")
out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
ops.Stream(record=False).gemm(ops.gemm_params(M, N, ops.linear_segs([(x, K, K)]), w, out, N, bias=b, tile=tile))
torch.cuda.synchronize()
ref = x.float() @ w.float().t() + b
err = (out.float() - ref).abs()
bad = err > 0.05
print("bad elements", int(bad.sum()), "of", M * N)
rows = bad.any(dim=1).nonzero().flatten()
cols = bad.any(dim=0).nonzero().flatten()
print("bad rows", rows[:40].tolist(), "...", len(rows))
print("bad cols", cols[:60].tolist(), "...", len(cols))
if len(rows):
    r = int(rows[0]); c = int(bad[r].nonzero()[0])
    print("first bad", r, c, float(out[r, c]), float(ref[r, c]), "bias", float(b[c]), "no-bias", float(ref[r, c] - b[c]))
if len(rows):
    r = int(rows[0])
    print("row", r, "out ", [round(float(v), 3) for v in out[r, :24]])
    print("row", r, "ref ", [round(float(v), 3) for v in ref[r, :24]])
    out2 = torch.zeros_like(out)
    ops.Stream(record=False).gemm(ops.gemm_params(M, N, ops.linear_segs([(x, K, K)]), w, out2, N, bias=b, tile=tile))
    torch.cuda.synchronize()
    bad2 = (out2.float() - ref).abs() > 0.05
    print("second run bad", int(bad2.sum()), "same set", bool((bad2 == bad).all()))
    out3 = torch.zeros_like(out)
    ops.Stream(record=False).gemm(ops.gemm_params(M, N, ops.linear_segs([(x, K, K)]), w, out3, N, tile=tile))
    torch.cuda.synchronize()
    bad3 = (out3.float() - (ref - b)).abs() > 0.05
    print("no-bias run bad", int(bad3.sum()))

"""ablate-8 check of the persistent epilogue's store path: every 16-byte unit must hold (i, 0, 0, 0) as uint32."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videomv_amd import _lib as L, ops
M, N, K, tile = (int(v) for v in sys.argv[1:5])
x = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
for rep in range(3):
    out = torch.full((M, N), -1.0, device="cuda", dtype=torch.bfloat16)
    ops.Stream(record=False).gemm(ops.gemm_params(M, N, ops.linear_segs([(x, K, K)]), w, out, N, tile=tile))
    torch.cuda.synchronize()
    u = out.view(torch.int16).view(M, N // 8, 8).to(torch.int32) & 0xffff
    first = u[:, :, 0] + (u[:, :, 1] << 16)
    rest = u[:, :, 2:].abs().sum(dim=2)
    rows = torch.arange(M, device="cuda")
    exp_i = ((rows % 48) // 16)[:, None].expand_as(first)
    bad_first = (first != exp_i)
    bad_rest = rest != 0
    print("rep", rep, "bad first", int(bad_first.sum()), "bad rest", int(bad_rest.sum()))
    if bad_first.any():
        idx = bad_first.nonzero()[:8]
        print([(int(a), int(b), hex(int(first[a, b]))) for a, b in idx])

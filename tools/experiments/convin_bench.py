#!/usr/bin/env python
"""The VAE encoder's conv_in (3 -> 128 channels, input padded to 8 channels: K = 72) over 48 x 256 x 256 rows, per tile family (GPU only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videomv_amd import _lib as L, ops
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gemm_bench import bench

BF = L.elem()
S = ops.Stream(record=False)
n, H, W, Cp, N = 48, 256, 256, 8, 128
M = n * H * W
x = torch.randn(M, Cp, device="cuda").to(BF)
w = torch.randn(N, 9 * Cp, device="cuda").to(BF)
b = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=BF)
for tile in [int(t) for t in sys.argv[1:]] or [0, 1, 2, 3, 5, 7, 9, 13, 22]:
    p = ops.gemm_params(M, N, ops.conv3x3_segs([(x, Cp, Cp)]), w, out, N, bias=b, geom=ops.Geom(OH=H, OW=W, IH=H, IW=W), tile=tile)
    try:
        S.gemm(p, "t")
    except Exception as e:
        print(f"tile {tile:2d}: {type(e).__name__}")
        continue
    ms = bench(lambda: S.gemm(p, "t"))
    print(f"tile {tile:2d}: {1000 * ms:8.1f} us   {M * N * 2 / ms / 1e9:7.1f} GB/s written")

#!/usr/bin/env python
"""Micro-benchmark of the implicit-GEMM kernels on the UNet's characteristic shapes (GPU only).
   python tools/gemm_bench.py [tile ...]   — prints TFLOP/s per (shape, tile)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videomv_amd import _lib as L, ops

BF = torch.bfloat16
def bench(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def main():
    tiles = [int(t) for t in sys.argv[1:]] or [L.TILE_128x160, L.TILE_256x160]
    S = ops.Stream(record=False)
    dev = "cuda"
    shapes = [  # (name, M, N, C, kind)
        ("lin  L0 N320 K320", 122880, 320, 320, "lin"), ("qkv  L0 N960 K320", 122880, 960, 320, "lin"),
        ("down L0 N320 K1280", 122880, 320, 1280, "lin"), ("conv L0 320->320", 122880, 320, 320, "conv"),
        ("tcnv L0 320", 122880, 320, 320, "tconv"), ("conv L1 640->640", 30720, 640, 640, "conv"),
        ("lin  L1 N640 K640", 30720, 640, 640, "lin"), ("conv L2 1280", 7680, 1280, 1280, "conv"),
    ]
    for name, M, N, C, kind in shapes:
        x = torch.randn(M, C, device=dev).to(BF)
        if kind == "lin":
            K = C; segs = ops.linear_segs([(x, C, C)]); geom = None
        elif kind == "conv":
            K = 9 * C; segs = ops.conv3x3_segs([(x, C, C)])
            hw = {122880: (40, 64), 30720: (20, 32), 7680: (10, 16)}[M]
            geom = ops.Geom(OH=hw[0], OW=hw[1], IH=hw[0], IW=hw[1])
        else:
            K = 3 * C; segs = ops.temporal_segs(x, C, C); geom = ops.Geom(F=24, P=M // 48)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF)
        b = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=BF)
        line = f"{name:22s}"
        for tile in tiles:
            if tile in (L.TILE_128x160, L.TILE_256x160) and N % 160: 
                line += "      -   "; continue
            p = ops.gemm_params(M, N, segs, w, out, N, bias=b, geom=geom, tile=tile)
            ms = bench(lambda: S.gemm(p))
            line += f" t{tile}:{2.0 * M * N * K / ms / 1e9:7.1f}"
        print(line, flush=True)

if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""GPU: the K = 1280 linears of the third level (folded LayerNorm, GEGLU, residual) on the one-block-per-CU wide tile (VMV_TILE_X256x256 /
the plan's 256 x 160 tile) against the 256-thread two-blocks-per-CU form (VMV_TILE_Y256x128, gemm_xglds.hip WNV = 4)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videomv_amd import _lib as L, ops

BF = L.elem()
lib = ops.Stream(record=False).lib
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return 1000.0 * e0.elapsed_time(e1) / reps


for name, M, N, K, kind, base in (("qkv L2 (LN)", 7680, 3840, 1280, "ln", L.TILE_X256x256), ("q L2 (LN)", 7680, 1280, 1280, "ln", L.TILE_X256x256),
                                  ("geglu L2 (LN)", 7680, 10240, 1280, "lngeglu", L.TILE_X256x256),
                                  ("attn.out L2", 7680, 1280, 1280, "res", L.TILE_256x160), ("ff.down L2", 7680, 1280, 5120, "res", L.TILE_256x160),
                                  ("attn.out L3", 1920, 1280, 1280, "res", 0), ("geglu mid (LN)", 1920, 10240, 1280, "lngeglu", L.TILE_P256x128),
                                  ("ff.down L1", 30720, 640, 2560, "res", L.TILE_X256x320), ("qkv L1 plain", 30720, 1920, 640, "plain", L.TILE_X256x256)):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = (torch.randn(M, K, generator=g, device="cuda") * 1.3 + 0.4).to(BF)
    w = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).to(BF)
    b = torch.randn(N, generator=g, device="cuda")
    No = N // 2 if kind == "lngeglu" else N
    out = torch.zeros(M, No, dtype=BF, device="cuda")
    kw = dict(bias=b)
    if kind in ("ln", "lngeglu"):
        kw.update(colsum=torch.randn(N, generator=g, device="cuda"), rowstat=torch.ones(M, 2, device="cuda"))
    if kind == "lngeglu":
        kw.update(epilogue=L.EPI_GEGLU)
    if kind == "res":
        kw.update(residual=torch.randn(M, No, generator=g, device="cuda").to(BF), ldr=No)
    line = f"{name:16s} {M}x{N}x{K}"
    for tile in (base, L.TILE_Y256x128):
        p = ops.gemm_params(M, N, ops.linear_segs([(x, K, K)]), w, out, No, tile=tile, **kw)
        if lib.vmv_gemm_validate(C.byref(p)) != 0:
            line += f" | tile {tile:2d} refused"; continue
        t = timeit(lambda: lib.vmv_gemm(C.byref(p), stream))
        line += f" | tile {tile:2d}: {t:7.1f} us {2.0 * M * N * K / t / 1e6:7.1f} TF"
    print(line, flush=True)

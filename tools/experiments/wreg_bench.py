#!/usr/bin/env python
"""GPU: the wide-wave register-staged kernel (VMV_TILE_W256x256, csrc/gemm_wreg.hip) against the plan's tile choice on the long plain
linears — the K = 1280 shapes where hipBLASLt leads (profiles/r4_vendor_yardstick.tsv: qkv L2 69.1 us, GEGLU-up L2 177.7 us bare)."""
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


print(f"{'shape':34s} {'plan tile':>10s} {'us':>8s} {'TFLOP/s':>8s} | {'W256x256 us':>12s} {'TFLOP/s':>8s} {'rel-l2':>9s}")
for name, M, N, K, kind, tile in (("qkv L2 (LN)", 7680, 3840, 1280, "ln", L.TILE_X256x256), ("geglu L2 (LN)", 7680, 10240, 1280, "lngeglu", L.TILE_X256x256),
                                  ("plain 7680x3840x1280", 7680, 3840, 1280, "plain", L.TILE_X256x256), ("plain 7680x10240x1280", 7680, 10240, 1280, "plain", L.TILE_X256x256),
                                  ("attn.out L2 (+res)", 7680, 1280, 1280, "res", L.TILE_256x160), ("ff.down L2 (+res)", 7680, 1280, 5120, "res", L.TILE_256x160),
                                  ("qkv L1 (LN)", 30720, 1920, 640, "ln", L.TILE_X256x256), ("geglu L1 (LN)", 30720, 5120, 640, "lngeglu", L.TILE_X256x256),
                                  ("ff.down L0 (+res)", 122880, 320, 1280, "res", L.TILE_P256x160), ("ff.down L1 (+res)", 30720, 640, 2560, "res", L.TILE_P256x160),
                                  ("geglu mid (LN)", 1920, 10240, 1280, "lngeglu", L.TILE_P256x128)):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = (torch.randn(M, K, generator=g, device="cuda") * 1.3 + 0.4).to(BF)
    w = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).to(BF)
    b = torch.randn(N, generator=g, device="cuda")
    No = N // 2 if kind == "lngeglu" else N
    o1, o2 = torch.zeros(M, No, dtype=BF, device="cuda"), torch.zeros(M, No, dtype=BF, device="cuda")
    kw = dict(bias=b)
    if kind in ("ln", "lngeglu"):
        xs = x.float()
        st = torch.stack([xs.mean(1), (xs.var(1, unbiased=False) + 1e-5).rsqrt()], dim=1).contiguous()
        kw.update(colsum=w.float().sum(1).contiguous(), rowstat=st)
    if kind == "lngeglu":
        kw.update(epilogue=L.EPI_GEGLU)
    if kind == "res":
        kw.update(residual=torch.randn(M, No, generator=g, device="cuda").to(BF), ldr=No)
    segs = ops.linear_segs([(x, K, K)])
    pa = ops.gemm_params(M, N, segs, w, o1, No, tile=tile, **kw)
    pb = ops.gemm_params(M, N, segs, w, o2, No, tile=L.TILE_W256x256, **kw)
    if lib.vmv_gemm_validate(C.byref(pa)) != 0:
        pa.tile = 0
    ta = timeit(lambda: lib.vmv_gemm(C.byref(pa), stream))
    if lib.vmv_gemm_validate(C.byref(pb)) != 0:
        print(f"{name:34s} W256x256 refused"); continue
    tb = timeit(lambda: lib.vmv_gemm(C.byref(pb), stream))
    fl = 2.0 * M * N * K
    err = float((o2.float() - o1.float()).norm() / o1.float().norm())
    print(f"{name + f' {M}x{N}x{K}':34s} {lib.vmv_gemm_pick_tile(C.byref(pa)):10d} {ta:8.1f} {fl / ta / 1e6:8.1f} | {tb:12.1f} {fl / tb / 1e6:8.1f} {err:9.2e}", flush=True)

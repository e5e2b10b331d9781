#!/usr/bin/env python
"""GPU: the plan's K = 1280 (third / fourth level) GEMMs under the dispatcher's own tile choice — run once with VMV_XGLDS_GM=1
VMV_GLDS_GM=1 (round-5 tile order) and once without (grouped order: gm row tiles x 32 / gm column tiles share an XCD's L2)."""
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


tag = f"XGLDS_GM={os.environ.get('VMV_XGLDS_GM', 'auto')} GLDS_GM={os.environ.get('VMV_GLDS_GM', 'auto')}"
print(f"# {tag}")
for name, M, N, K, kind, tile, ks in (("qkv L2 (LN)", 7680, 3840, 1280, "ln", L.TILE_X256x256, 0), ("geglu L2 (LN)", 7680, 10240, 1280, "lngeglu", L.TILE_X256x256, 0),
                                      ("attn.out L2", 7680, 1280, 1280, "res", L.TILE_256x160, 0), ("ff.down L2", 7680, 1280, 5120, "res", L.TILE_256x160, 0),
                                      ("tconv L2", 7680, 1280, 3840, "tconv", L.TILE_256x160, 0), ("tconv L3 ks3", 1920, 1280, 3840, "tconv", L.TILE_256x128, 3),
                                      ("qkv L1 (plain tile)", 30720, 1920, 640, "ln", L.TILE_X256x256, 0), ("ff.down L1", 30720, 640, 2560, "res", L.TILE_P256x160, 0),
                                      ("geglu mid (LN)", 1920, 10240, 1280, "lngeglu", L.TILE_P256x128, 0)):
    g = torch.Generator(device="cuda").manual_seed(1)
    Kx = K // 3 if kind == "tconv" else K
    x = (torch.randn(M, Kx, generator=g, device="cuda") * 1.3 + 0.4).to(BF)
    w = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).to(BF)
    b = torch.randn(N, generator=g, device="cuda")
    No = N // 2 if kind == "lngeglu" else N
    out = torch.zeros(M, No, dtype=BF, device="cuda")
    kw = dict(bias=b)
    segs, geom = ops.linear_segs([(x, K, K)]), None
    if kind in ("ln", "lngeglu"):
        kw.update(colsum=torch.randn(N, generator=g, device="cuda"), rowstat=torch.ones(M, 2, device="cuda"))
    if kind == "lngeglu":
        kw.update(epilogue=L.EPI_GEGLU)
    if kind == "res":
        kw.update(residual=torch.randn(M, No, generator=g, device="cuda").to(BF), ldr=No)
    if kind == "tconv":
        segs, geom = ops.temporal_segs(x, Kx, Kx), ops.Geom(F=24, P=M // 48)
    ws = torch.zeros(max(1, ks) * M * N, device="cuda") if ks > 1 else None
    p = ops.gemm_params(M, N, segs, w, out, No, geom=geom, tile=tile, ksplit=ks, workspace=ws, **kw)
    if lib.vmv_gemm_validate(C.byref(p)) != 0:
        print(f"{name:22s} tile {tile} refused"); continue
    t = timeit(lambda: lib.vmv_gemm(C.byref(p), stream))
    print(f"{name:22s} {M}x{N}x{K} tile {tile:2d} ks{ks}: {t:8.1f} us {2.0 * M * N * K / t / 1e6:8.1f} TFLOP/s", flush=True)

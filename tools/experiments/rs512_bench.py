#!/usr/bin/env python
"""GPU: the row-stationary kernel at K = 512 (round 6: RsCfg<2, 16>) against the tile kernels on the init TemporalTransformer's linears
(8 heads x 64 = 512 channels on the 320-channel level; one CFG branch's rows: shared prefix)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videomv_amd import _lib as L, ops, packing as P

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


CAND = [L.TILE_P256x128, L.TILE_P256x160, L.TILE_Q128x128, L.TILE_X256x320, L.TILE_X256x256, L.TILE_256x128, L.TILE_256x160, L.TILE_G128x128, L.TILE_G128x160]
print(f"{'shape':34s} {'RS us':>8s} {'TFLOP/s':>8s} | best tile kernel")
for M in (61440, 24576, 15360):
    for tag, N, K, kind in (("qkv (LN)", 1536, 512, "ln"), ("geglu (LN)", 4096, 512, "lngeglu"), ("attn.out (+res)", 512, 512, "res"), ("proj_out (+res)", 320, 512, "res"),
                            ("proj_in", 512, 320, "plain")):
        g = torch.Generator(device="cuda").manual_seed(1)
        x = (torch.randn(M, K, generator=g, device="cuda") * 1.3 + 0.4).to(BF)
        w = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).to(BF)
        b = torch.randn(N, generator=g, device="cuda")
        No = N // 2 if kind == "lngeglu" else N
        out = torch.zeros(M, No, dtype=BF, device="cuda")
        res = torch.randn(M, No, generator=g, device="cuda").to(BF)
        kw = dict(bias=b)
        if kind in ("ln", "lngeglu"):
            kw.update(colsum=torch.randn(N, generator=g, device="cuda"), ln_eps=1e-5)
        if kind == "lngeglu":
            kw.update(epilogue=L.EPI_GEGLU)
        if kind == "res":
            kw.update(residual=res, ldr=No)
        st = torch.zeros(M, 2, device="cuda") + 1.0

        def mk(tile):
            kk = dict(kw)
            if tile not in (L.TILE_RS, L.TILE_RS256) and "ln_eps" in kk:      # the tile kernels take the statistics from a pass of their own
                kk.pop("ln_eps"); kk["rowstat"] = st
            return ops.gemm_params(M, N, ops.linear_segs([(x, K, K)]), w, out, No, tile=tile, **kk)
        fl = 2.0 * M * N * K
        p_rs = mk(L.TILE_RS256)
        t_rs = timeit(lambda: lib.vmv_gemm(C.byref(p_rs), stream)) if lib.vmv_gemm_validate(C.byref(p_rs)) == 0 else float("nan")
        best = None
        for t in CAND:
            p_t = mk(t)
            if lib.vmv_gemm_validate(C.byref(p_t)) != 0:
                continue
            tt = timeit(lambda: lib.vmv_gemm(C.byref(p_t), stream), reps=10)
            if best is None or tt < best[0]:
                best = (tt, t)
        p_a = mk(0)
        print(f"{tag + f' {M}x{N}x{K}':34s} {t_rs:8.1f} {fl / t_rs / 1e6:8.1f} | tile {best[1]:2d} {best[0]:7.1f} us   policy picks {lib.vmv_gemm_pick_tile(C.byref(p_a))}", flush=True)

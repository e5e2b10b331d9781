#!/usr/bin/env python
"""GPU: the frame-resident temporal convolution (VMV_TILE_TFR, csrc/gemm_tfr.hip) against the tile kernels on the UNet's shapes.
Per shape: plain convolution on TFR vs the policy's tile kernel; the folded form (statistics + table + TFR with gn_table) vs the
three-launch form (statistics + apply + tile kernel) — the comparison that decides the default."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videomv_amd import _lib as L, ops, packing as P

BF = L.elem()
S = ops.Stream(record=False)
lib = S.lib
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)


REPS = int(os.environ.get("TFR_BENCH_REPS", "20"))
ONLY = os.environ.get("TFR_BENCH_ONLY", "")          # substring filter on the shape tag (profiling runs)


def timeit(fn, reps=None, warm=3):
    reps = reps or REPS
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return 1000.0 * e0.elapsed_time(e1) / reps


print(f"{'shape':28s} {'tile':>5s} {'plain us':>9s} {'TFLOP/s':>8s} | {'tfr us':>8s} {'TFLOP/s':>8s} | {'3-launch us':>11s} {'folded us':>10s} {'eff TFLOP/s':>11s}")
for tag, Bn, F_, Pp, Cc in (("L0 40x64", 2, 24, 2560, 320), ("L0 40x64 B=1", 1, 24, 2560, 320), ("L0 32x32", 2, 24, 1024, 320), ("L1 40x64", 2, 24, 640, 640),
                            ("L1 32x32", 2, 24, 256, 640), ("L2 40x64", 2, 24, 160, 1280), ("rank0of8 L0", 1, 24, 320, 320)):
    if ONLY and ONLY not in tag:
        continue
    M, N, rps = Bn * F_ * Pp, Cc, F_ * Pp
    g = torch.Generator(device="cuda").manual_seed(1)
    x = (torch.randn(M, Cc, generator=g, device="cuda") * 1.3 + 0.4).to(BF)
    w = (torch.randn(N, 3 * Cc, generator=g, device="cuda") * (3 * Cc) ** -0.5).to(BF)
    b = torch.randn(N, generator=g, device="cuda")
    res = torch.randn(M, N, generator=g, device="cuda").to(BF)
    gamma, beta = torch.ones(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
    y, o1, o2 = torch.zeros(M, Cc, dtype=BF, device="cuda"), torch.zeros(M, N, dtype=BF, device="cuda"), torch.zeros(M, N, dtype=BF, device="cuda")
    tab = torch.zeros(Bn * 2 * Cc, device="cuda")
    ws = torch.zeros(ops.gn_partial_floats(M, rps, Cc) + 64, device="cuda")
    tot = torch.zeros(2, Bn * ops.GN_TOT, dtype=torch.int64, device="cuda")
    geom = ops.Geom(F=F_, P=Pp)
    fl = 2.0 * M * N * 3 * Cc

    def gp(src, out, tile, **kw):
        return ops.gemm_params(M, N, ops.temporal_segs(src, Cc, Cc), w, out, N, bias=b, geom=geom, tile=tile, **kw)
    p_pol = gp(y, o1, 0)
    os.environ["VMV_GEMM_TFR"] = "0"
    pol = lib.vmv_gemm_pick_tile(C.byref(p_pol))        # (the policy's tile without the new kernel: the env is read once per process, so force it)
    p_pol.tile = pol if pol != L.TILE_TFR else L.TILE_X256x320
    p_tfr = gp(y, o2, L.TILE_TFR)
    ok = lib.vmv_gemm_validate(C.byref(p_tfr)) == 0
    t_pol = timeit(lambda: lib.vmv_gemm(C.byref(p_pol), stream))
    t_tfr = timeit(lambda: lib.vmv_gemm(C.byref(p_tfr), stream)) if ok else float("nan")
    gn_a = ops.gn_params(x, Cc, Cc, M, rps, ws, gamma, beta, 1e-5, True, y, Cc, totals=tot[0], totals_clear=tot[1], clear_count=Bn * ops.GN_TOT)
    gn_t = ops.gn_params(x, Cc, Cc, M, rps, ws, gamma, beta, 1e-5, False, tab, Cc, totals=tot[0], totals_clear=tot[1], clear_count=Bn * ops.GN_TOT)
    p_fold = gp(x, o2, L.TILE_TFR, gn_table=tab, gn_rows_per_stat=rps, gn_silu=True)

    def three():
        tot[0].zero_()
        lib.vmv_groupnorm_stats(C.byref(gn_a), stream); lib.vmv_groupnorm_apply(C.byref(gn_a), stream); lib.vmv_gemm(C.byref(p_pol), stream)

    def folded():
        tot[0].zero_()
        lib.vmv_groupnorm_stats(C.byref(gn_t), stream); lib.vmv_groupnorm_table(C.byref(gn_t), stream); lib.vmv_gemm(C.byref(p_fold), stream)
    t3 = timeit(three)
    tf = timeit(folded) if ok else float("nan")
    if ok:
        three(); folded(); torch.cuda.synchronize()
        err = float((o1.float() - o2.float()).norm() / o1.float().norm())
    else:
        err = float("nan")
    print(f"{tag:28s} {p_pol.tile:5d} {t_pol:9.1f} {fl / t_pol / 1e6:8.0f} | {t_tfr:8.1f} {fl / t_tfr / 1e6:8.0f} | {t3:11.1f} {tf:10.1f} {fl / tf / 1e6:11.0f}   rel-L2 folded vs 3-launch {err:.1e}", flush=True)

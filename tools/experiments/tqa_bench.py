#!/usr/bin/env python
"""GPU: the fused q | k | v projection + temporal attention (VMV_EPI_TATTN, csrc/gemm_tqa.hip) against the plan's two launches
(row-stationary folded-LayerNorm q | k | v GEMM + attn_short_kernel) on the TemporalTransformer shapes of the K = 320 level."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videomv_amd import _lib as L, ops, packing as P

BF = L.elem()
S = ops.Stream(record=False)
lib = S.lib
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
REPS = int(os.environ.get("TQA_BENCH_REPS", "20"))


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


print(f"{'shape':24s} {'qkv us':>8s} {'attn us':>8s} {'two us':>8s} | {'fused us':>9s} {'TFLOP/s':>8s} {'rel-l2 vs two':>13s}")
for tag, nb, F_, Pp, heads in (("L0 40x64 B=2", 2, 24, 2560, 5), ("L0 40x64 B=1", 1, 24, 2560, 5), ("L0 32x32 B=2", 2, 24, 1024, 5),
                               ("L0 48x48 B=2", 2, 24, 2304, 5), ("rank0of8 L0 40x64", 2, 24, 320, 5)):
    K, inner = 320, 64 * heads
    M = nb * F_ * Pp
    g = torch.Generator(device="cuda").manual_seed(1)
    x = (torch.randn(M, K, generator=g, device="cuda") * 1.3 + 0.4).to(BF)
    w = torch.randn(3 * inner, K, generator=g, device="cuda") * (1.5 * K ** -0.5)
    gamma, beta = 1 + 0.1 * torch.randn(K, generator=g, device="cuda"), 0.1 * torch.randn(K, generator=g, device="cuda")
    wf, bf, cs = P.fold_layernorm(w, None, gamma, beta)
    wf, bf, cs = wf.contiguous(), bf.contiguous(), cs.contiguous()
    whm, bhm, chm = (P.qkv_head_major(t).contiguous() for t in (wf, bf, cs))
    qkv = torch.zeros(M, 3 * inner, dtype=BF, device="cuda")
    two, one = torch.zeros(M, inner, dtype=BF, device="cuda"), torch.zeros(M, inner, dtype=BF, device="cuda")
    p_qkv = ops.gemm_params(M, 3 * inner, ops.linear_segs([(x, K, K)]), wf, qkv, 3 * inner, bias=bf, colsum=cs, ln_eps=1e-5)
    mp = lambda ld: ops.seq_map(F_ * Pp * ld, ld, Pp * ld, inner=Pp)
    base = qkv.data_ptr()
    p_att = ops.attn_params(base, base + 2 * inner, base + 4 * inner, two, mp(3 * inner), mp(3 * inner), mp(3 * inner), mp(inner), nb * Pp, heads, F_, F_, 0.125)
    p_one = ops.gemm_params(M, 3 * inner, ops.linear_segs([(x, K, K)]), whm, one, inner, bias=bhm, colsum=chm, ln_eps=1e-5, epilogue=L.EPI_TATTN,
                            epi_scale=0.125, geom=ops.Geom(F=F_, P=Pp))
    assert lib.vmv_gemm_validate(C.byref(p_one)) == 0
    t_q = timeit(lambda: lib.vmv_gemm(C.byref(p_qkv), stream))
    t_a = timeit(lambda: lib.vmv_attention(C.byref(p_att), stream))
    t_2 = timeit(lambda: (lib.vmv_gemm(C.byref(p_qkv), stream), lib.vmv_attention(C.byref(p_att), stream)))
    t_1 = timeit(lambda: lib.vmv_gemm(C.byref(p_one), stream))
    fl = 2.0 * M * 3 * inner * K + 4.0 * M * F_ * 64 * heads
    err = float((one.float() - two.float()).norm() / two.float().norm())
    print(f"{tag:24s} {t_q:8.1f} {t_a:8.1f} {t_2:8.1f} | {t_1:9.1f} {fl / t_1 / 1e6:8.1f} {err:13.2e}", flush=True)

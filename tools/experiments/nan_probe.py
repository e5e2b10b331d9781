#!/usr/bin/env python
"""Where does a NaN / inf in the input of each kernel family go?  (round 4: fp16 stores saturate through MODE.FP16_OVFL; the finite
check at the end of decode / the sampling loop relies on true NaN / inf surviving the kernels.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videomv_amd import _lib as L, ops

BF = L.elem()
S = ops.Stream(record=False)
dev = "cuda"


def report(name, t):
    t = t.float()
    print(f"{name:34s} finite={bool(torch.isfinite(t).all())} nan={int(torch.isnan(t).sum())} inf={int(torch.isinf(t).sum())} max|finite|={float(t[torch.isfinite(t)].abs().max()) if torch.isfinite(t).any() else 0:.4g}")


for bad in (float("nan"), float("inf")):
    print("==== bad value:", bad)
    M, N, K = 256, 128, 128
    a = torch.randn(M, K, device=dev).to(BF); a[3, 5] = bad
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF)
    for tile in (0, L.TILE_64x64, L.TILE_128x128, L.TILE_G128x128, L.TILE_256x128):
        o = torch.zeros(M, N, device=dev, dtype=BF)
        S.gemm(ops.gemm_params(M, N, ops.linear_segs([(a, K, K)]), w, o, N, tile=tile)); torch.cuda.synchronize()
        report(f"gemm tile {tile} row3", o[3]); 
    o32 = torch.zeros(M, N, device=dev)
    S.gemm(ops.gemm_params(M, N, ops.linear_segs([(a, K, K)]), w, o32, N, out_fp32=True)); torch.cuda.synchronize()
    report("gemm fp32 out row3", o32[3])
    # GroupNorm: fused one-launch, and statistics + apply
    rows, C = 64, 64
    x = torch.randn(rows, C, device=dev).to(BF); x[2, 3] = bad
    y = torch.zeros(rows, C, device=dev, dtype=BF)
    gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    ws = torch.zeros(1 << 16, device=dev)
    p = ops.gn_params(x, C, C, rows, rows, ws, gam, bet, 1e-5, True, y, C)
    S.groupnorm_fused(p, ops.gn_fused_cols(rows, C) or C); torch.cuda.synchronize()
    report("groupnorm fused", y)
    y.zero_()
    p = ops.gn_params(x, C, C, rows, rows, ws, gam, bet, 1e-5, True, y, C)
    S.groupnorm_stats(p); S.groupnorm_apply(p); torch.cuda.synchronize()
    report("groupnorm stats+apply", y)
    tot = torch.zeros(2, ops.GN_TOT, dtype=torch.int64, device=dev)
    y.zero_()
    p = ops.gn_params(x, C, C, rows, rows, ws, gam, bet, 1e-5, True, y, C, totals=tot[0], totals_clear=tot[1], clear_count=ops.GN_TOT)
    S.groupnorm_stats(p); S.groupnorm_apply(p); torch.cuda.synchronize()
    report("groupnorm totals+apply", y)
    # LayerNorm
    yl = torch.zeros(rows, C, device=dev, dtype=BF)
    S.layernorm(ops.ln_params(x, C, yl, C, gam, bet, rows, C, 1e-5)); torch.cuda.synchronize()
    report("layernorm row2", yl[2])
    # attention (one problem, 64 queries / keys)
    q = torch.randn(64, 64, device=dev).to(BF); k = torch.randn(64, 64, device=dev).to(BF); v = torch.randn(64, 64, device=dev).to(BF)
    q[1, 2] = bad
    o = torch.zeros(64, 64, device=dev, dtype=BF)
    m = ops.seq_map(64 * 64, 0, 64, inner=1)
    S.attention(ops.attn_params(q, k, v, o, m, m, m, m, 1, 1, 64, 64, 0.125)); torch.cuda.synchronize()
    report("attention q-row1", o[1])
    v2 = v.clone(); v2[4, 4] = bad; q2 = torch.randn(64, 64, device=dev).to(BF)
    S.attention(ops.attn_params(q2, k, v2, o, m, m, m, m, 1, 1, 64, 64, 0.125)); torch.cuda.synchronize()
    report("attention bad V", o)

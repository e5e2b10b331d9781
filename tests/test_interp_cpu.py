"""`not gpu`: the plan interpreter used as the GPU tests' reference is itself checked against torch.nn.functional
(conv2d incl. stride-2 / nearest-x2 / channel concat + fused 1x1 skip, conv3d (3,1,1), scaled-dot-product
attention with the three index maps, GEGLU) — this pins the *contract* of include/vmv.h and the weight packing."""
import pytest
import torch
import torch.nn.functional as Fn

from videomv_amd import _lib as L
from videomv_amd import ops, packing as P
from tests import plan_interp as I

BF = L.elem()


def g(seed):
    return torch.Generator().manual_seed(seed)


def close(a, b, tol=1e-2):
    a, b = a.float(), b.float()
    assert float((a - b).abs().max()) <= tol * float(b.abs().max().clamp_min(1e-6)), float((a - b).abs().max())


@pytest.mark.parametrize("stride,ups", [(1, 0), (2, 0), (1, 1)])
def test_conv3x3_contract(stride, ups):
    n, IH, IW, C0, C1, N = 2, 6, 5, 16, 8, 12
    OH = (IH + 1) // 2 if stride == 2 else (IH * 2 if ups else IH)
    OW = (IW + 1) // 2 if stride == 2 else (IW * 2 if ups else IW)
    Cin = C0 + C1
    wt = (torch.randn(N, Cin, 3, 3, generator=g(2)) * 0.1).to(BF).float()
    ws = (torch.randn(N, Cin, 1, 1, generator=g(7)) * 0.1).to(BF).float()
    x0 = torch.randn(n * IH * IW, C0, generator=g(1)).to(BF)
    x1 = torch.randn(n * IH * IW, C1, generator=g(4)).to(BF)
    wp = torch.cat([wt.permute(0, 2, 3, 1).reshape(N, -1), ws.reshape(N, Cin)], dim=1).to(BF).contiguous()
    b = torch.randn(N, generator=g(3))
    out = torch.zeros(n * OH * OW, N)
    srcs = [(x0, C0, C0), (x1, C1, C1)]
    segs = ops.conv3x3_segs(srcs) + (ops.linear_segs(srcs) if (stride == 1 and not ups) else [])
    if not (stride == 1 and not ups):
        wp = wp[:, : 9 * Cin].contiguous()
    I.gemm(ops.gemm_params(n * OH * OW, N, segs, wp, out, N, bias=b, out_fp32=True,
                           geom=ops.Geom(OH=OH, OW=OW, IH=IH, IW=IW, stride=stride, ups=ups)))
    x = torch.cat([x0, x1], dim=1).float().view(n, IH, IW, Cin).permute(0, 3, 1, 2)
    img = Fn.interpolate(x, scale_factor=2, mode="nearest") if ups else x
    ref = Fn.conv2d(img, wt, b, stride=stride, padding=1)
    if stride == 1 and not ups:
        ref = ref + Fn.conv2d(x, ws)
    close(out, ref.permute(0, 2, 3, 1).reshape(-1, N), 1e-5)


def test_pack_conv3x3_pads_input_channels():
    w = torch.randn(8, 4, 3, 3, generator=g(1))
    p = P.pack_conv3x3(w, "cpu")
    assert p.shape == (8, 72)
    assert torch.equal(p.view(8, 9, 8)[:, :, 4:], torch.zeros(8, 9, 4, dtype=BF))
    assert torch.equal(p.view(8, 3, 3, 8)[:, 2, 0, :4], w[:, :, 2, 0].to(BF))


def test_temporal_contract():
    Bn, F_, Pp, Cc = 2, 4, 6, 8
    M = Bn * F_ * Pp
    wt = (torch.randn(Cc, Cc, 3, 1, 1, generator=g(2)) * 0.2).to(BF).float()
    x = torch.randn(M, Cc, generator=g(1)).to(BF)
    b = torch.randn(Cc, generator=g(3))
    out = torch.zeros(M, Cc)
    wp = P.pack_tconv(wt, "cpu")      # keep alive: the argument block only holds raw pointers
    I.gemm(ops.gemm_params(M, Cc, ops.temporal_segs(x, Cc, Cc), wp, out, Cc, bias=b, out_fp32=True,
                           geom=ops.Geom(F=F_, P=Pp)))
    x5 = x.float().view(Bn, F_, Pp, Cc).permute(0, 3, 1, 2)[..., None]
    ref = Fn.conv3d(x5, wt, b, padding=(1, 0, 0))[..., 0].permute(0, 2, 3, 1).reshape(M, Cc)
    close(out, ref, 1e-5)


def test_geglu_contract():
    M, I2, K = 10, 64, 16
    w = (torch.randn(I2, K, generator=g(2)) * 0.3).to(BF)
    b = torch.randn(I2, generator=g(3))
    a = torch.randn(M, K, generator=g(1)).to(BF)
    out = torch.zeros(M, I2 // 2)
    wi, bi = P.geglu_interleave(w).contiguous(), P.geglu_interleave(b).contiguous()
    I.gemm(ops.gemm_params(M, I2, ops.linear_segs([(a, K, K)]), wi, out, I2 // 2, bias=bi, epilogue=L.EPI_GEGLU,
                           out_fp32=True))
    h = a.float() @ w.float().t() + b
    x, gate = h.chunk(2, dim=-1)
    close(out, x * Fn.gelu(gate), 1e-5)


def test_attention_maps_contract():
    B, F_, HW, heads = 2, 3, 5, 2
    inner = heads * 64
    T = B * F_ * HW
    qkv = torch.randn(T, 3 * inner, generator=g(1)).to(BF)
    q5 = qkv.float().view(B, F_, HW, 3, heads, 64)
    sc = 64 ** -0.5
    # spatial
    o = torch.zeros(T, inner, dtype=BF)
    mp = lambda ld: ops.seq_map(HW * ld, 0, ld, inner=1)
    base = qkv.data_ptr()
    I.attention(ops.attn_params(base, base + 2 * inner, base + 4 * inner, o, mp(3 * inner), mp(3 * inner), mp(3 * inner),
                                mp(inner), B * F_, heads, HW, HW, sc))
    q, k, v = (q5[:, :, :, i].permute(0, 1, 3, 2, 4) for i in range(3))          # b f h n d
    ref = Fn.scaled_dot_product_attention(q, k, v).permute(0, 1, 3, 2, 4).reshape(T, inner)
    close(o, ref)
    # temporal: sequences run over f at fixed (b, pixel)
    o2 = torch.zeros(T, inner, dtype=BF)
    mt = lambda ld: ops.seq_map(F_ * HW * ld, ld, HW * ld, inner=HW)
    I.attention(ops.attn_params(base, base + 2 * inner, base + 4 * inner, o2, mt(3 * inner), mt(3 * inner), mt(3 * inner),
                                mt(inner), B * HW, heads, F_, F_, sc))
    q, k, v = (q5[:, :, :, i].permute(0, 2, 3, 1, 4) for i in range(3))          # b p h f d
    ref = Fn.scaled_dot_product_attention(q, k, v).permute(0, 3, 1, 2, 4).reshape(T, inner)
    close(o2, ref)


def test_fused_qkv_temporal_attention_argument_block_equals_the_two_launch_form():
    """VMV_EPI_TATTN (vmv.h, csrc/gemm_tqa.hip): ONE GEMM argument block with the head-major weight [head][q | k | v][64] (packing.
    qkv_head_major), geometry (F, P) and the softmax scale against the plan's two launches — folded-LayerNorm q | k | v GEMM, then the
    temporal attention through its index maps.  Same interpreter arithmetic on both sides: equal to rounding."""
    from videomv_amd import packing as P
    nb, F_, Pp, heads, K = 2, 6, 7, 3, 320
    inner, M = 64 * heads, nb * F_ * Pp
    x = (torch.randn(M, K, generator=g(3)) * 1.3 + 0.5).to(BF)
    w = torch.randn(3 * inner, K, generator=g(4)) * K ** -0.5
    gamma, beta = 1 + 0.2 * torch.randn(K, generator=g(5)), 0.2 * torch.randn(K, generator=g(6))
    wf, bf, cs = P.fold_layernorm(w, None, gamma, beta)
    wf, bf, cs = wf.contiguous(), bf.contiguous(), cs.contiguous()
    qkv = torch.zeros(M, 3 * inner, dtype=BF)
    I.gemm(ops.gemm_params(M, 3 * inner, ops.linear_segs([(x, K, K)]), wf, qkv, 3 * inner, bias=bf, colsum=cs, ln_eps=1e-5))
    two = torch.zeros(M, inner, dtype=BF)
    mt = lambda ld: ops.seq_map(F_ * Pp * ld, ld, Pp * ld, inner=Pp)
    base = qkv.data_ptr()
    I.attention(ops.attn_params(base, base + 2 * inner, base + 4 * inner, two, mt(3 * inner), mt(3 * inner), mt(3 * inner), mt(inner),
                                nb * Pp, heads, F_, F_, 0.125))
    whm, bhm, chm = (P.qkv_head_major(t).contiguous() for t in (wf, bf, cs))
    # the permutation: row 192 h + 64 s + d of the head-major matrix is row s * inner + 64 h + d of [q | k | v]
    assert torch.equal(whm[192 * 1 + 64 * 2 + 5], wf[2 * inner + 64 * 1 + 5]) and torch.equal(bhm[192 * 2 + 7], bf[64 * 2 + 7])
    one = torch.full((M, inner + 8), 3.0, dtype=BF)
    I.gemm(ops.gemm_params(M, 3 * inner, ops.linear_segs([(x, K, K)]), whm, one, inner + 8, bias=bhm, colsum=chm, ln_eps=1e-5,
                           epilogue=L.EPI_TATTN, epi_scale=0.125, geom=ops.Geom(F=F_, P=Pp)))
    close(one[:, :inner], two.float(), 2e-2)
    assert bool((one[:, inner:].float() == 3.0).all())


def test_ff_fused_argument_block_equals_the_two_gemm_form():
    """The fused FeedForward's argument block (interleaved W1, K-permuted W2: packing.ff_down_permute) through the interpreter
    against the two-GEMM form (LayerNorm-folded GEGLU GEMM with in-kernel statistics, then the down projection with the
    residual) on the same weights: same rounding points, so equal to accumulation order."""
    from videomv_amd import _lib as L, ops, packing as P
    torch.manual_seed(0)
    M, Cc = 96, 320
    BF = L.elem()
    x = (torch.randn(M, Cc) * 1.5 + 2.0).to(BF)
    w1 = torch.randn(8 * Cc, Cc) * Cc ** -0.5
    b1 = torch.randn(8 * Cc)
    w2 = torch.randn(Cc, 4 * Cc) * (4 * Cc) ** -0.5
    b2 = torch.randn(Cc)
    gamma, beta = 1 + 0.2 * torch.randn(Cc), 0.2 * torch.randn(Cc)
    wf, bf, cs = P.fold_layernorm(w1, b1, gamma, beta)
    w1p, b1p, csp = P.geglu_interleave(wf).contiguous(), P.geglu_interleave(bf).contiguous(), P.geglu_interleave(cs).contiguous()
    w2n, w2p = w2.to(BF).contiguous(), P.ff_down_permute(w2).to(BF).contiguous()
    hid = torch.zeros(M, 4 * Cc, dtype=BF)
    out_a, out_b = torch.zeros(M, Cc, dtype=BF), torch.zeros(M, Cc, dtype=BF)
    I.gemm(ops.gemm_params(M, 8 * Cc, ops.linear_segs([(x, Cc, Cc)]), w1p, hid, 4 * Cc, bias=b1p, colsum=csp, ln_eps=1e-5,
                                     epilogue=L.EPI_GEGLU))
    I.gemm(ops.gemm_params(M, Cc, ops.linear_segs([(hid, 4 * Cc, 4 * Cc)]), w2n, out_a, Cc, bias=b2, residual=x, ldr=Cc))
    I.ff_fused(ops.ff_params(M, Cc, x, Cc, w1p, b1p, w2p, b2, out_b, Cc, residual=x, ldr=Cc, ln_eps=1e-5))
    err = float((out_a.float() - out_b.float()).norm() / out_a.float().norm())
    assert err < 2e-3, err


def test_groupnorm_fold_argument_block_equals_apply_then_gemm():
    """vmv_groupnorm_table + VmvGemmParams.gn_table (the GroupNorm -> proj_in fold of the transformers, gemm_rs.hip): the interpreter's
    reading of the two argument blocks gives bitwise what stats + apply + plain GEMM gives — the contract the GPU test holds the
    kernels to."""
    nstat, rps, K, N = 3, 32, 64, 48
    M = nstat * rps
    x = (torch.randn(M, K, generator=g(1)) * 1.5 + 0.3).to(BF)
    gamma, beta = 1 + 0.2 * torch.randn(K, generator=g(2)), 0.1 * torch.randn(K, generator=g(3))
    w, b = (torch.randn(N, K, generator=g(4)) * K ** -0.5).to(BF), torch.randn(N, generator=g(5))
    part = torch.zeros(ops.gn_partial_floats(M, rps, K) + 64)
    y, tab = torch.zeros(M, K, dtype=BF), torch.zeros(nstat, 2, K)
    o_ref, o_new = torch.zeros(M, N, dtype=BF), torch.zeros(M, N, dtype=BF)
    gp = ops.gn_params(x, K, K, M, rps, part, gamma, beta, 1e-6, False, y, K)
    I.groupnorm_stats(gp); I.groupnorm(gp)
    I.gemm(ops.gemm_params(M, N, ops.linear_segs([(y, K, K)]), w, o_ref, N, bias=b))
    gt = ops.gn_params(x, K, K, M, rps, part, gamma, beta, 1e-6, False, tab, K)
    I.groupnorm_stats(gt); I.groupnorm_table(gt)
    I.gemm(ops.gemm_params(M, N, ops.linear_segs([(x, K, K)]), w, o_new, N, bias=b, gn_table=tab, gn_rows_per_stat=rps))
    ref = torch.nn.functional.group_norm(x.float().view(nstat, rps, K).permute(0, 2, 1), 32, gamma, beta, 1e-6).permute(0, 2, 1).reshape(M, K)
    close(y, ref, 1e-2)
    assert float((o_new.float() - o_ref.float()).abs().max()) <= 2 ** -6 * float(o_ref.float().abs().max())      # (one rounding of x_hat apart at most)


def test_temporal_conv_groupnorm_fold_argument_block():
    """VmvGemmParams.gn_table + gn_silu on three TEMPORAL segments (the frame-resident kernel, csrc/gemm_tfr.hip): the interpreter's
    reading — norm + SiLU applied to the SOURCE rows, the zero frames -1 / F padded afterwards — equals statistics + apply(silu) + the
    plain temporal convolution, and F.group_norm + F.silu + Conv3d; the table is per (sample, segment channel), not per ktot."""
    Bn, F_, Pp, Cc, N = 2, 4, 6, 64, 32
    M, rps = Bn * F_ * Pp, F_ * Pp
    x = (torch.randn(M, Cc, generator=g(1)) * 1.5 + 0.7).to(BF)
    gamma, beta = 1 + 0.2 * torch.randn(Cc, generator=g(2)), 0.3 * torch.randn(Cc, generator=g(3))
    wt = torch.randn(N, Cc, 3, 1, 1, generator=g(4)) * (3 * Cc) ** -0.5
    w, b = P.pack_tconv(wt, "cpu"), torch.randn(N, generator=g(5))
    part = torch.zeros(ops.gn_partial_floats(M, rps, Cc) + 64)
    y, tab = torch.zeros(M, Cc, dtype=BF), torch.zeros(Bn, 2, Cc)
    o_ref, o_new = torch.zeros(M, N, dtype=BF), torch.zeros(M, N, dtype=BF)
    gp = ops.gn_params(x, Cc, Cc, M, rps, part, gamma, beta, 1e-5, True, y, Cc)
    I.groupnorm_stats(gp); I.groupnorm(gp)
    I.gemm(ops.gemm_params(M, N, ops.temporal_segs(y, Cc, Cc), w, o_ref, N, bias=b, geom=ops.Geom(F=F_, P=Pp)))
    gt = ops.gn_params(x, Cc, Cc, M, rps, part, gamma, beta, 1e-5, False, tab, Cc)
    I.groupnorm_stats(gt); I.groupnorm_table(gt)
    I.gemm(ops.gemm_params(M, N, ops.temporal_segs(x, Cc, Cc), w, o_new, N, bias=b, geom=ops.Geom(F=F_, P=Pp), gn_table=tab,
                           gn_rows_per_stat=rps, gn_silu=True))
    assert float((o_new.float() - o_ref.float()).abs().max()) <= 2 ** -6 * float(o_ref.float().abs().max())
    xn = torch.nn.functional.silu(torch.nn.functional.group_norm(x.float().view(Bn, rps, Cc).permute(0, 2, 1), 32, gamma, beta, 1e-5))
    x5 = xn.permute(0, 2, 1).reshape(Bn, F_, Pp, Cc).permute(0, 3, 1, 2)[..., None]
    ref = torch.nn.functional.conv3d(x5, wt.to(BF).float(), b, padding=(1, 0, 0))[..., 0].permute(0, 2, 3, 1).reshape(M, N)
    close(o_new, ref, 2e-2)

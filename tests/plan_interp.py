"""TEST INFRASTRUCTURE: a CPU interpreter of the C-ABI argument blocks (include/vmv.h), written from the header's
contract in plain torch.  It lets the `not gpu` suite execute a recorded UNet plan on host memory and compare it
with the oracle, so that the host logic (segment lists, weight packing, index maps, buffer reuse) is verified
without a GPU.  It is never imported by the product package and says nothing about the HIP kernels themselves —
those are checked on the GPU against torch references and against the oracle.
"""
import ctypes as C
import math

import numpy as np
import torch

from videomv_amd import _lib as L


def _view(ptr, n, kind):
    """Torch view of n elements at raw host address ptr. kind: 'elem' (the 16-bit type in use) | 'f32'."""
    if kind == "elem":
        arr = np.ctypeslib.as_array((C.c_uint16 * n).from_address(ptr))
        return torch.from_numpy(arr.view(np.int16)).view(L.elem())
    if kind == "i64":
        return torch.from_numpy(np.ctypeslib.as_array((C.c_int64 * n).from_address(ptr)))
    arr = np.ctypeslib.as_array((C.c_float * n).from_address(ptr))
    return torch.from_numpy(arr)


def _rows(ptr, nrows, ld, kind="elem"):
    return _view(ptr, nrows * ld, kind).view(nrows, ld)


def gemm(p: L.GemmParams):
    M, N = p.M, p.N
    m = torch.arange(M)
    cols = []
    tfr_gn = bool(p.gn_table) and p.nseg == 3 and p.seg[0].mode == L.SEG_TEMPORAL      # GroupNorm (+ SiLU) folded into a temporal conv (vmv.h gn_silu)
    for s in range(p.nseg):
        sg = p.seg[s]
        if sg.mode == L.SEG_LINEAR:
            src_row = m.clone()
            valid = torch.ones(M, dtype=torch.bool)
        elif sg.mode == L.SEG_SPATIAL:
            hw = p.OH * p.OW
            n = m // hw
            rem = m % hw
            oy, ox = rem // p.OW, rem % p.OW
            iy = oy * p.stride + sg.d0
            ix = ox * p.stride + sg.d1
            VH, VW = p.IH << p.ups, p.IW << p.ups
            valid = (iy >= 0) & (iy < VH) & (ix >= 0) & (ix < VW)
            src_row = n * p.IH * p.IW + (iy.clamp(0, VH - 1) >> p.ups) * p.IW + (ix.clamp(0, VW - 1) >> p.ups)
        else:
            f = (m // p.P) % p.F
            valid = (f + sg.d0 >= 0) & (f + sg.d0 < p.F)
            src_row = (m + sg.d0 * p.P).clamp(0, M - 1)
        nsrc = int(src_row.max()) + 1
        src = _rows(sg.src, nsrc, sg.ld)[:, : sg.k].float()
        a = src[src_row]
        if tfr_gn:            # the norm is applied to the SOURCE rows; the frame-axis zero padding comes after it (Conv3d pads the normalised tensor)
            nstat = (M + p.gn_rows_per_stat - 1) // p.gn_rows_per_stat
            tab = _view(p.gn_table, nstat * 2 * sg.k, "f32").view(nstat, 2, sg.k)
            st = src_row // p.gn_rows_per_stat
            a = a * tab[st, 0] + tab[st, 1]
            if p.gn_silu:
                a = torch.nn.functional.silu(a)
            a = a.to(L.elem()).float()
        a[~valid] = 0
        cols.append(a)
    A = torch.cat(cols, dim=1)
    if p.gn_table and not tfr_gn:            # GroupNorm folded into the A rows: elem(x * scale + shift) per (stat group, channel)
        nstat = (M + p.gn_rows_per_stat - 1) // p.gn_rows_per_stat
        tab = _view(p.gn_table, nstat * 2 * p.ktot, "f32").view(nstat, 2, p.ktot)
        st = m // p.gn_rows_per_stat
        A = (A * tab[st, 0] + tab[st, 1]).to(L.elem()).float()
    if p.wgroup_rows > 0:     # grouped weights (vmv.h): rows [g R, (g + 1) R) multiply the matrix at W + g * stride elements
        R = p.wgroup_rows
        acc = torch.empty(M, N)
        for gi in range((M + R - 1) // R):
            Wg = _rows(p.W + 2 * gi * p.wgroup_stride, N, p.ktot).float()
            acc[gi * R:(gi + 1) * R] = A[gi * R:(gi + 1) * R] @ Wg.t()
    else:
        W = _rows(p.W, N, p.ktot).float()
        acc = A @ W.t()
    if p.rowstat:          # LayerNorm folded into the GEMM: rstd[m] * (acc - mean[m] * colsum[n])
        st = _view(p.rowstat, 2 * M, "f32").view(M, 2)
        acc = (acc - st[:, :1] * _view(p.colsum, N, "f32")[None, :]) * st[:, 1:2]
    elif p.colsum and p.ln_eps > 0:      # the same, statistics taken from the rows the GEMM multiplies (VmvGemmParams.ln_eps)
        mean = A.mean(dim=1, keepdim=True)
        rstd = torch.rsqrt(((A * A).mean(dim=1, keepdim=True) - mean * mean).clamp_min(0) + p.ln_eps)
        acc = (acc - mean * _view(p.colsum, N, "f32")[None, :]) * rstd
    if p.bias:
        acc = acc + _view(p.bias, N, "f32")
    if p.epilogue == L.EPI_TATTN:      # fused q | k | v (head-major W rows) + attention over the F frames of every (sample, pixel, head)
        heads = N // 192
        nb = M // (p.F * p.P)
        qkv = acc.to(L.elem()).float().view(nb, p.F, p.P, heads, 3, 64)
        q, k, v = (qkv[..., i, :].permute(0, 2, 3, 1, 4) for i in range(3))            # [nb, P, heads, F, 64]
        att = torch.softmax((q @ k.transpose(-1, -2)) * p.epi_scale, dim=-1)
        acc = (att @ v).permute(0, 3, 1, 2, 4).reshape(M, heads * 64)
    if p.epilogue == L.EPI_GEGLU:
        acc = acc.view(M, N // 32, 2, 16)
        x, g = acc[:, :, 0], acc[:, :, 1]
        acc = (x * torch.nn.functional.gelu(g)).reshape(M, N // 2)
    No = acc.shape[1]
    if p.rowvec:
        ngroups = (M + p.rowvec_div - 1) // p.rowvec_div
        rv = _rows(p.rowvec, ngroups, p.rowvec_ld, "f32")[:, :No]
        acc = acc + rv[m // p.rowvec_div]
    if p.act == L.ACT_SILU:
        acc = torch.nn.functional.silu(acc)
    elif p.act == L.ACT_GELU:
        acc = torch.nn.functional.gelu(acc)
    if p.residual:
        acc = acc + (p.res_scale if p.res_scale != 0.0 else 1.0) * _rows(p.residual, M, p.ldr)[:, :No].float()
    out = _rows(p.out, M, p.ldo, "f32" if p.out_fp32 else "elem")
    out[:, :No] = acc if p.out_fp32 else acc.to(L.elem())


def _gn_input(p):
    Cc = p.C0 + p.C1
    x = _rows(p.x, p.rows, p.ld)[:, : p.C0].float()
    if p.C1:
        x = torch.cat([x, _rows(p.x1, p.rows, p.ld1)[:, : p.C1].float()], dim=1)
    return x, Cc


GN_REC, GN_NREP = 8, 8          # int64 per record of VmvGroupNormParams.totals, replicas per (stat, group) (include/vmv.h)


def _gn_pilots(x, nstat, rps, Cc):
    """pilot[stat][group] = first element of the group: first row of the stat group, first channel of the group."""
    return x.view(nstat, rps, 32, Cc // 32)[:, 0, :, 0].float().contiguous()


def groupnorm_stats(p: L.GroupNormParams):
    """partial[stat][chunk][group] = (sum, sum of squares) of x - pilot over chunk_rows rows x C/32 channels."""
    x, Cc = _gn_input(p)
    nstat = p.rows // p.rows_per_stat
    nchunk = (p.rows_per_stat + p.chunk_rows - 1) // p.chunk_rows
    part = _view(p.partial, nstat * nchunk * 64, "f32").view(nstat, nchunk, 32, 2)
    pil = _gn_pilots(x, nstat, p.rows_per_stat, Cc)
    xg = x.view(nstat, p.rows_per_stat, 32, Cc // 32) - pil[:, None, :, None]
    for c in range(nchunk):
        blk = xg[:, c * p.chunk_rows:(c + 1) * p.chunk_rows]
        part[:, c, :, 0] = blk.sum(dim=(1, 3))
        part[:, c, :, 1] = (blk * blk).sum(dim=(1, 3))
    if p.totals:       # two-limb fixed-point integer accumulation (order-independent), as the kernel does
        tot = _view(p.totals, nstat * 32 * GN_NREP * GN_REC, "i64").view(nstat, 32, GN_NREP, GN_REC)
        v = part.float()                                             # [nstat][nchunk][32][2]
        hi = torch.floor(v)
        lo = ((v - hi) * float(2 ** 40)).to(torch.int64)
        for c in range(nchunk):                                      # chunk c -> replica c % NREP
            tot[:, :, c % GN_NREP, 0:2] += hi[:, c].to(torch.int64)
            tot[:, :, c % GN_NREP, 2:4] += lo[:, c]
        tot[:, :, 0, 4] = pil.view(torch.int32).to(torch.int64) & 0xffffffff


def groupnorm_table(p: L.GroupNormParams):
    groupnorm(p, table=True)


def groupnorm(p: L.GroupNormParams, table=False):
    """apply: fold the partial sums (of `fold_ranks` gathered shards when > 1), normalise, affine, optional SiLU."""
    x, Cc = _gn_input(p)
    nstat = p.rows // p.rows_per_stat
    nchunk = (p.rows_per_stat + p.chunk_rows - 1) // p.chunk_rows
    R = max(1, p.fold_ranks)
    nr = float(p.rows_per_stat) * (Cc // 32)
    if p.totals:
        rec = _view(p.totals, R * nstat * 32 * GN_NREP * GN_REC, "i64").view(R, nstat, 32, GN_NREP, GN_REC)
        lim = rec[..., 0:4].sum(dim=3)                                                        # exact integer fold of the replicas
        sr = lim[..., 0:2].double() + lim[..., 2:4].double() / float(2 ** 40)            # [R][nstat][32][2]
        pil = (rec[..., 0, 4] & 0xffffffff).to(torch.int32).view(torch.float32).double()   # [R][nstat][32]
        d = pil - pil[0:1]
        S = (sr[..., 0] + nr * d).sum(dim=0)
        Q = (sr[..., 1] + 2.0 * d * sr[..., 0] + nr * d * d).sum(dim=0)
        n = nr * R
        m = S / n
        mean = pil[0] + m
        var = (Q / n - m * m).clamp_min(0.0)
        if p.totals_clear:
            _view(p.totals_clear, p.clear_count, "i64").zero_()
    else:
        assert R == 1, "shards are folded through the totals records"
        part = _view(p.partial, nstat * nchunk * 64, "f32").view(nstat, nchunk, 32, 2).double().sum(dim=1)
        m = part[..., 0] / nr
        mean = _gn_pilots(x, nstat, p.rows_per_stat, Cc).double() + m
        var = (part[..., 1] / nr - m * m).clamp_min(0.0)
    if table:          # vmv_groupnorm_table: y is the fp32 [nstat][2][C] scale / shift table
        gamma, beta = _view(p.gamma, Cc, "f32"), _view(p.beta, Cc, "f32")
        rstd = torch.rsqrt(var.float() + p.eps)                                  # [nstat][32]
        sc = rstd[:, :, None] * gamma.view(1, 32, Cc // 32)
        sh = beta.view(1, 32, Cc // 32) - mean.float()[:, :, None] * sc
        tab = _view(p.y, nstat * 2 * Cc, "f32").view(nstat, 2, Cc)
        tab[:, 0], tab[:, 1] = sc.reshape(nstat, Cc), sh.reshape(nstat, Cc)
        return
    xg = x.view(nstat, p.rows_per_stat, 32, Cc // 32)
    y = ((xg - mean.float()[:, None, :, None]) * torch.rsqrt(var.float() + p.eps)[:, None, :, None]).view(p.rows, Cc)
    y = y * _view(p.gamma, Cc, "f32") + _view(p.beta, Cc, "f32")
    if p.silu:
        y = torch.nn.functional.silu(y)
    _rows(p.y, p.rows, p.ldy)[:, :Cc] = y.to(L.elem())


def groupnorm_fused(p: L.GroupNormParams):
    """vmv_groupnorm_fused: statistics (two-pass) + apply in one op."""
    x, Cc = _gn_input(p)
    nstat = p.rows // p.rows_per_stat
    xg = x.view(nstat, p.rows_per_stat, 32, Cc // 32).double()
    mean = xg.mean(dim=(1, 3), keepdim=True)
    var = ((xg - mean) ** 2).mean(dim=(1, 3), keepdim=True)
    y = ((xg - mean) * torch.rsqrt(var + p.eps)).float().view(p.rows, Cc)
    y = y * _view(p.gamma, Cc, "f32") + _view(p.beta, Cc, "f32")
    if p.silu:
        y = torch.nn.functional.silu(y)
    _rows(p.y, p.rows, p.ldy)[:, :Cc] = y.to(L.elem())


def permute_copy(p: L.CopyParams):
    """dst[i0][i1][i2][:] = src[i0*ss0 + i1*ss1 + i2*ss2 + :], 16-byte units (8 bf16)."""
    span = (p.n0 - 1) * p.ss0 + (p.n1 - 1) * p.ss1 + (p.n2 - 1) * p.ss2 + p.inner16
    src = _view(p.src, span * 8, "elem").view(torch.int16)
    dst = _view(p.dst, p.n0 * p.n1 * p.n2 * p.inner16 * 8, "elem").view(torch.int16)
    v = torch.as_strided(src, (p.n0, p.n1, p.n2, p.inner16 * 8), (p.ss0 * 8, p.ss1 * 8, p.ss2 * 8, 1))
    dst.view(p.n0, p.n1, p.n2, p.inner16 * 8).copy_(v)


def layernorm(p: L.LayerNormParams):
    x = _rows(p.x, p.rows, p.ldx)[:, : p.C].float()
    if p.stats_out:        # statistics only: (mean, rstd) per row
        mean = x.mean(dim=1)
        rstd = torch.rsqrt(x.var(dim=1, unbiased=False) + p.eps)
        _view(p.stats_out, 2 * p.rows, "f32").view(p.rows, 2).copy_(torch.stack([mean, rstd], dim=1))
        return
    y = torch.nn.functional.layer_norm(x, (p.C,), _view(p.gamma, p.C, "f32"), _view(p.beta, p.C, "f32"), p.eps)
    _rows(p.y, p.rows, p.ldy)[:, : p.C] = y.to(L.elem())


def _seq_rows(ptr, mp, o, h, n, hd=64):
    base = (o // mp.inner) * mp.s_outer + (o % mp.inner) * mp.s_inner + h * hd
    flat = _view(ptr, int(base + (n - 1) * mp.s_row + hd), "elem")
    idx = base + torch.arange(n)[:, None] * mp.s_row + torch.arange(hd)[None, :]
    return flat, idx


def attention(p: L.AttnParams):
    hd = p.head_dim or 64
    for o in range(p.n_outer):
        for h in range(p.heads):
            qf, qi = _seq_rows(p.q, p.qm, o, h, p.Nq, hd)
            kf, ki = _seq_rows(p.k, p.km, o // p.kv_div, h, p.Nk, hd)
            vf, vi = _seq_rows(p.v, p.vm, o // p.kv_div, h, p.Nk, hd)
            q, k, v = qf[qi].float(), kf[ki].float(), vf[vi].float()
            s = q @ k.t() * p.scale
            if p.causal:
                s = s.masked_fill(torch.ones_like(s, dtype=torch.bool).triu(1), float("-inf"))
            out = (torch.softmax(s, dim=-1) @ v).to(L.elem())
            of, oi = _seq_rows(p.o, p.om, o, h, p.Nq, hd)
            of[oi.reshape(-1)] = out.reshape(-1)


def softmax_rows(p: L.SoftmaxParams):
    s = _rows(p.s, p.rows, p.lds, "f32")[:, : p.n]
    _rows(p.p, p.rows, p.ldp)[:, : p.n] = torch.softmax(s * p.scale, dim=-1).to(L.elem())


def ff_fused(p: L.FfParams):
    """vmv_ff_fused: out = residual + W2 (x' * gelu(gate')) + b2 with (x' | gate') = W1 LN(x) + b1, hidden rounded to elem; W1 rows
    interleaved x | gate in 16-row blocks, W2's K axis permuted per 32-channel block (packing.FF_DOWN_ORDER)."""
    from videomv_amd.packing import FF_DOWN_ORDER
    M, Cc = p.M, p.C
    x = _rows(p.x, M, p.ldx)[:, :Cc].float()
    if p.ln_eps > 0:
        mean = x.mean(dim=1, keepdim=True)
        x = ((x - mean) * torch.rsqrt(((x - mean) ** 2).mean(dim=1, keepdim=True) + p.ln_eps)).to(L.elem()).float()
    h = x @ _rows(p.w1, 8 * Cc, Cc).float().t()
    if p.b1:
        h = h + _view(p.b1, 8 * Cc, "f32")
    h = h.view(M, 8 * Cc // 32, 2, 16)
    h = (h[:, :, 0] * torch.nn.functional.gelu(h[:, :, 1])).reshape(M, 4 * Cc).to(L.elem()).float()
    idx = torch.tensor(FF_DOWN_ORDER, dtype=torch.long)
    hp = h.view(M, 4 * Cc // 32, 32)[:, :, idx].reshape(M, 4 * Cc)          # natural -> the order W2's K axis is stored in
    acc = hp @ _rows(p.w2, Cc, 4 * Cc).float().t()
    if p.b2:
        acc = acc + _view(p.b2, Cc, "f32")
    if p.residual:
        acc = acc + _rows(p.residual, M, p.ldr)[:, :Cc].float()
    _rows(p.out, M, p.ldo)[:, :Cc] = acc.to(L.elem())


def comm_sim(p: L.CommParams):
    """VMV_OP_COMM on a SIMULATED communicator (vmv_comm_create_sim: the only kind a CPU test can hold): all-to-all recv = send,
    all-gather recv[j] = send for every j."""
    lib = L.load()
    assert lib.vmv_comm_is_sim(p.comm) == 1
    W, n = lib.vmv_comm_world(p.comm), int(p.bytes)
    src = torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * (n * (W if p.kind == L.COMM_ALL_TO_ALL else 1))).from_address(p.send)))
    dst = torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * (n * W)).from_address(p.recv)))
    if p.kind == L.COMM_ALL_TO_ALL:
        dst.copy_(src)
    else:
        dst.view(W, n).copy_(src.view(1, n).expand(W, n))


def run_recorded(recorded):
    for op, params in recorded:
        if op == L.OP_GEMM:
            gemm(params)
        elif op == L.OP_GN_STATS:
            groupnorm_stats(params)
        elif op == L.OP_GN_APPLY:
            groupnorm(params)
        elif op == L.OP_LAYERNORM:
            layernorm(params)
        elif op == L.OP_ATTENTION:
            attention(params)
        elif op == L.OP_SOFTMAX:
            softmax_rows(params)
        elif op == L.OP_COPY:
            permute_copy(params)
        elif op == L.OP_GN_FUSED:
            groupnorm_fused(params)
        elif op == L.OP_FF:
            ff_fused(params)
        elif op == L.OP_GN_TABLE:
            groupnorm_table(params)
        elif op == L.OP_COMM:
            comm_sim(params)
        else:
            raise ValueError(op)


# ---- direct glue kernels (same contracts as videomv_amd/ops.py's launchers)
def latent_to_rows(x, rows, Cpad, nrep):
    nb, Cc, F_, H, W = x.shape
    r = x.permute(0, 2, 3, 4, 1).reshape(nb * F_ * H * W, Cc)
    full = torch.zeros(nb * F_ * H * W, Cpad)
    full[:, :Cc] = r
    rows.copy_(full.repeat(nrep, 1).to(L.elem()))


def latent_to_rows_keep(x, rows, ld, nrep):
    nb, Cc, F_, H, W = x.shape
    r = x.permute(0, 2, 3, 4, 1).reshape(nb * F_ * H * W, Cc).to(L.elem())
    rows.view(nrep, nb * F_ * H * W, ld)[:, :, :Cc] = r[None]


def i2v_temporal_adapter(inp, ld_in, out_ptr, ld_out, w, F_, HW, nrep, scale):
    import torch.nn.functional as Fn
    x = _rows(_p(inp), F_ * HW, ld_in)[:, :4].float().view(F_, HW, 4).permute(1, 0, 2)        # pix f c
    ln_g, ln_b, Wqkv, Wo, bo = w[0:4], w[4:8], w[8:104].view(24, 4), w[104:136].view(4, 8), w[136:140]
    W1, b1, W2, b2 = w[140:204].view(16, 4), w[204:220], w[220:284].view(4, 16), w[284:288]
    hn = Fn.layer_norm(x, (4,), ln_g, ln_b, 1e-5)
    q, k, v = (hn @ Wqkv.t()).chunk(3, dim=-1)
    sp = lambda t: t.reshape(HW, F_, 2, 4).permute(0, 2, 1, 3)
    a = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) * 0.5, dim=-1) @ sp(v)
    y = a.permute(0, 2, 1, 3).reshape(HW, F_, 8) @ Wo.t() + bo + x
    z = Fn.gelu(y @ W1.t() + b1) @ W2.t() + b2 + y
    res = (scale * z).permute(1, 0, 2).reshape(F_ * HW, 4).to(L.elem())
    out = _view(_p(out_ptr), (nrep * F_ * HW - 1) * ld_out + 4, "elem")
    idx = torch.arange(nrep * F_ * HW)[:, None] * ld_out + torch.arange(4)[None, :]
    out[idx.reshape(-1)] = res.repeat(nrep, 1).reshape(-1)


def adaptive_avgpool_rows(inp, ld, out, ldo, n, Cc, IH, IW, OH, OW):
    import torch.nn.functional as Fn
    x = _rows(_p(inp), n * IH * IW, ld)[:, :Cc].float().view(n, IH, IW, Cc).permute(0, 3, 1, 2)
    y = Fn.adaptive_avg_pool2d(x, (OH, OW)).permute(0, 2, 3, 1).reshape(n * OH * OW, Cc)
    _rows(_p(out), n * OH * OW, ldo)[:, :Cc] = y.to(L.elem())


def posterior_sample(moments_rows, ld, noise, z, scale):
    n, zc, H, W = z.shape
    m = moments_rows.view(n, H * W, ld)
    mean = m[:, :, :zc].permute(0, 2, 1).reshape(n, zc, H, W)
    logvar = m[:, :, zc:2 * zc].permute(0, 2, 1).reshape(n, zc, H, W).clamp(-30.0, 20.0)
    z.copy_(scale * (mean + torch.exp(0.5 * logvar) * noise))


def lgm_x0_views(eps_rows, ld, branch, xt, idx4, c_recip, c_recipm1, inv_scale, out):
    _, Cc, F_, H, W = xt.shape
    e = eps_rows.view(-1, ld)[branch * F_ * H * W:(branch + 1) * F_ * H * W, :Cc].view(F_, H * W, Cc)
    for v, f in enumerate(idx4):
        out[v] = inv_scale * (c_recip * xt[0, :, f] - c_recipm1 * e[f].t().reshape(Cc, H, W))


def lgm_pack_input(decoded, rays, out):
    x = (decoded * 0.5 + 0.5).clamp(0, 1)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    out[:, 0:3] = (x - mean) / std
    out[:, 3:9] = rays


def lgm_render_to_vae(images, out):
    out.copy_((torch.nn.functional.interpolate(images, size=out.shape[-2:], mode="nearest") - 0.5) / 0.5)


def ddim_x0_step(x0_cond, x0_uncond, xt, guide, c_recip, c_recipm1, a_prev, clamp=None, sigma=0.0, noise=None):
    x0 = x0_uncond + guide * (x0_cond - x0_uncond)
    if clamp:
        x0 = x0.clamp(-clamp, clamp)
    eps = (c_recip * xt - x0) / c_recipm1
    nx = math.sqrt(a_prev) * x0 + math.sqrt(1 - a_prev - sigma * sigma) * eps
    if sigma:
        nx = nx + sigma * noise
    xt.copy_(nx)


def gaussian_activation(raw, ld, out, n, workspace):
    x = raw.view(-1, ld)[:n, :14].float()
    rot = x[:, 7:11]
    rot = rot / rot.norm(dim=0, keepdim=True).clamp_min(1e-12)
    out.view(-1, 14)[:n] = torch.cat([x[:, 0:3].clamp(-1, 1), torch.sigmoid(x[:, 3:4]),
                                      0.1 * torch.nn.functional.softplus(x[:, 4:7]), rot, 0.5 * torch.tanh(x[:, 11:14]) + 0.5], dim=1)


def _p(t):
    return t if isinstance(t, int) else t.data_ptr()


def rows_to_nchw(rows, ld, out):
    n, Cc, H, W = out.shape
    out.copy_(rows.view(n, H * W, ld)[:, :, :Cc].float().permute(0, 2, 1).reshape(n, Cc, H, W))


def emb_combine_silu(temb, cam, out, rows, Cc, rows_per_t, cam_rows):
    r = torch.arange(rows)
    v = temb[r // rows_per_t]
    if cam is not None:
        v = v + cam[r % cam_rows]
    out.copy_(torch.nn.functional.silu(v).to(L.elem()))


def sinusoidal(t, out, n, dim):
    half = dim // 2
    freq = torch.pow(10000, -torch.arange(half).float() / half)
    s = torch.outer(t.float(), freq)
    out.copy_(torch.cat([torch.cos(s), torch.sin(s)], dim=1).to(L.elem()))


def cfg_ddim_step(eps_rows, ld, xt, guide_scale, c_recip, c_recipm1, c_sqrt_ac, c_sqrt_1mac, a_prev, v_pred=False,
                  x0_out=None, clamp=0.0, sigma=0.0, noise=None):
    _, Cc, F_, H, W = xt.shape
    FHW = F_ * H * W
    e = eps_rows.view(2, FHW, ld)[:, :, :Cc]
    y, u = e[0].t().reshape(1, Cc, F_, H, W), e[1].t().reshape(1, Cc, F_, H, W)
    out = u + guide_scale * (y - u)
    f = torch.float32
    x0 = (torch.tensor(c_sqrt_ac, dtype=f) * xt - torch.tensor(c_sqrt_1mac, dtype=f) * out) if v_pred else \
        (torch.tensor(c_recip, dtype=f) * xt - torch.tensor(c_recipm1, dtype=f) * out)
    if clamp:
        x0 = x0.clamp(-clamp, clamp)
    eps = (torch.tensor(c_recip, dtype=f) * xt - x0) / torch.tensor(c_recipm1, dtype=f)
    ap, sg = torch.tensor(a_prev, dtype=f), torch.tensor(float(sigma), dtype=f)
    nx = torch.sqrt(ap) * x0 + torch.sqrt(1 - ap - sg * sg) * eps
    if sigma:
        nx = nx + sg * noise
    xt.copy_(nx)
    if x0_out is not None:
        x0_out.copy_(x0)


def install(monkeypatch):
    """Route every launch of videomv_amd.ops to this interpreter (CPU tensors)."""
    from videomv_amd import ops

    def _go(self, op, params, fn, label):
        if self.record:
            self.labels.append(label)
            self.recorded.append((op, params))
            self.nops += 1
        else:
            run_recorded([(op, params)])

    def run(self, first=0, last=None):
        run_recorded(self.recorded[first:last])

    def init(self, record=False):
        self.lib = L.load()      # symbols only; nothing is launched
        self.record = record
        self.plan = None
        self.keep, self.nops, self.labels, self.recorded, self.graph = [], 0, [], [], None

    from videomv_amd import gs as _gs

    def render_cpu(self, gaussians, cam_view, cam_view_proj, cam_pos=None, bg_color=None, scale_modifier=1):
        from oracle.gs_ref import render_views
        bg = torch.ones(3) if bg_color is None else bg_color.float().cpu()
        imgs, alphas = zip(*[render_views(gaussians[b].float().cpu(), cam_view[b].float().cpu(), cam_view_proj[b].float().cpu(),
                                          self.size, self.fovy, bg) for b in range(gaussians.shape[0])])
        return {"image": torch.stack(imgs), "alpha": torch.stack(alphas)}

    monkeypatch.setattr(_gs.GaussianRenderer, "render", render_cpu)
    monkeypatch.setattr(ops.Stream, "__init__", init)
    monkeypatch.setattr(ops.Stream, "_go", _go)
    monkeypatch.setattr(ops.Stream, "run", run)
    for name in ("latent_to_rows", "latent_to_rows_keep", "rows_to_nchw", "emb_combine_silu", "sinusoidal", "cfg_ddim_step",
                 "i2v_temporal_adapter", "adaptive_avgpool_rows", "posterior_sample", "gaussian_activation", "lgm_x0_views", "lgm_pack_input", "lgm_render_to_vae",
                 "ddim_x0_step"):
        monkeypatch.setattr(ops, name, globals()[name])

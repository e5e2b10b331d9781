"""``AutoencoderKL`` — drop-in for the reference SD-VAE wrapper (tools/modules/autoencoder.py:31-104) registered under
``AUTO_ENCODER``; ``decode(z)`` runs on the gfx950 kernels (``Decoder.forward`` :654-687, ``ResnetBlock`` :316-336,
``AttnBlock`` :366-390, ``Upsample`` :456-460, ``Normalize`` = GroupNorm(32, eps 1e-6)).

State-dict names/shapes are the reference's (248 keys at full size, encoder included so that
``VQGAN_autoencoder.pth`` loads with ``strict=True`` after the ``first_stage_model.`` prefix filter, :65-74).
``encode`` / ``encode_firsr_stage`` (used by the I2VGen entrance for ``local_image`` and by the LGM refinement loop) run on
the same kernels (``VaeEncoderEngine``); the posterior noise is drawn on the host RNG like the reference.

Decoder plan (channels-last 16-bit rows, one chunk of n frames):
  post_quant 1x1 -> conv_in 3x3 -> Res -> Attn(1 head, d = C: GEMM QK^T -> row softmax -> GEMM PV per frame)
  -> Res -> 4 levels x 3 Res (1x1 nin_shortcut folded into conv2's K loop) with nearest-x2 folded into the
  up-sampling conv's gather -> GN + swish -> conv_out (fp32 rows) -> NCHW.
"""
import collections
import math
import os
from typing import Dict

import torch
import torch.nn as nn

from .registry import AUTO_ENCODER
from . import _lib as L
from . import ops
from . import packing as P
from .unet_engine import Pool, Act
from .unet_t2v import _Holder


def _res_shapes(p, ci, co):
    s = [(f"{p}.norm1.weight", (ci,)), (f"{p}.norm1.bias", (ci,)), (f"{p}.conv1.weight", (co, ci, 3, 3)),
         (f"{p}.conv1.bias", (co,)), (f"{p}.norm2.weight", (co,)), (f"{p}.norm2.bias", (co,)),
         (f"{p}.conv2.weight", (co, co, 3, 3)), (f"{p}.conv2.bias", (co,))]
    if ci != co:
        s += [(f"{p}.nin_shortcut.weight", (co, ci, 1, 1)), (f"{p}.nin_shortcut.bias", (co,))]
    return s


def _attn_shapes(p, c):
    s = [(f"{p}.norm.weight", (c,)), (f"{p}.norm.bias", (c,))]
    for n in ("q", "k", "v", "proj_out"):
        s += [(f"{p}.{n}.weight", (c, c, 1, 1)), (f"{p}.{n}.bias", (c,))]
    return s


def vae_param_shapes(dd: dict, embed_dim: int) -> Dict[str, tuple]:
    """Full AutoencoderKL manifest (encoder, decoder, quant convs); checked against tests/golden/manifest_vae_full.json."""
    ch, ch_mult, nrb = dd["ch"], list(dd["ch_mult"]), dd["num_res_blocks"]
    zc, in_ch, out_ch = dd["z_channels"], dd["in_channels"], dd["out_ch"]
    nres = len(ch_mult)
    s = [("encoder.conv_in.weight", (ch, in_ch, 3, 3)), ("encoder.conv_in.bias", (ch,))]
    in_mult = (1,) + tuple(ch_mult)
    block_in = ch
    for lvl in range(nres):
        block_in = ch * in_mult[lvl]
        block_out = ch * ch_mult[lvl]
        for i in range(nrb):
            s += _res_shapes(f"encoder.down.{lvl}.block.{i}", block_in, block_out)
            block_in = block_out
        if lvl != nres - 1:
            s += [(f"encoder.down.{lvl}.downsample.conv.weight", (block_in, block_in, 3, 3)),
                  (f"encoder.down.{lvl}.downsample.conv.bias", (block_in,))]
    s += _res_shapes("encoder.mid.block_1", block_in, block_in) + _attn_shapes("encoder.mid.attn_1", block_in)
    s += _res_shapes("encoder.mid.block_2", block_in, block_in)
    s += [("encoder.norm_out.weight", (block_in,)), ("encoder.norm_out.bias", (block_in,)),
          ("encoder.conv_out.weight", (2 * zc if dd.get("double_z", True) else zc, block_in, 3, 3)),
          ("encoder.conv_out.bias", (2 * zc if dd.get("double_z", True) else zc,))]
    block_in = ch * ch_mult[-1]
    s += [("decoder.conv_in.weight", (block_in, zc, 3, 3)), ("decoder.conv_in.bias", (block_in,))]
    s += _res_shapes("decoder.mid.block_1", block_in, block_in) + _attn_shapes("decoder.mid.attn_1", block_in)
    s += _res_shapes("decoder.mid.block_2", block_in, block_in)
    for lvl in reversed(range(nres)):
        block_out = ch * ch_mult[lvl]
        for i in range(nrb + 1):
            s += _res_shapes(f"decoder.up.{lvl}.block.{i}", block_in, block_out)
            block_in = block_out
        if lvl != 0:
            s += [(f"decoder.up.{lvl}.upsample.conv.weight", (block_in, block_in, 3, 3)),
                  (f"decoder.up.{lvl}.upsample.conv.bias", (block_in,))]
    s += [("decoder.norm_out.weight", (block_in,)), ("decoder.norm_out.bias", (block_in,)),
          ("decoder.conv_out.weight", (out_ch, block_in, 3, 3)), ("decoder.conv_out.bias", (out_ch,))]
    s += [("quant_conv.weight", (2 * embed_dim, 2 * zc, 1, 1)), ("quant_conv.bias", (2 * embed_dim,)),
          ("post_quant_conv.weight", (zc, embed_dim, 1, 1)), ("post_quant_conv.bias", (zc,))]
    return dict(s)


class _VaeEngine:
    """Shared pieces of the decoder / encoder plans: packing, GroupNorm(1e-6)+swish, ResnetBlock, 1-head AttnBlock."""

    def __init__(self, dd: dict, sd: Dict[str, torch.Tensor], n: int, h: int, w: int, device, packed=None):
        """packed: the .wt of another engine of the same kind, model and device — packed weights are immutable and
        shape-independent, so the engines of one model (other frame counts / resolutions) share ONE copy."""
        self.dd, self.n, self.h, self.w, self.device = dd, n, h, w, device
        self.pool = Pool(device)
        self.S = ops.Stream(record=True)
        self.S.tuner = ops.make_tuner(self)      # measured per-shape (tile, split-K) choices: videomv_amd/tuned_gemm.json
        self._keep = []
        self._gnws = torch.empty(4 << 20, dtype=torch.float32, device=device)
        if packed is not None:
            self.wt = packed
        else:
            self.wt = {}
            self._pack(sd)
            P.check_finite_weights(self.wt, type(self).__name__)
        self._build()

    def _packers(self, sd):
        dev, w = self.device, self.wt

        def conv(key, fold_skip=None):
            ww = sd[key + ".weight"]
            b = sd[key + ".bias"].float()
            if fold_skip is not None and (fold_skip + ".weight") in sd:
                wp = ww.permute(0, 2, 3, 1).reshape(ww.shape[0], -1)
                ws = sd[fold_skip + ".weight"].reshape(ww.shape[0], -1)
                w[key + ".weight"] = P.pack_linear(torch.cat([wp, ws], dim=1), dev)
                b = b + sd[fold_skip + ".bias"].float()
            else:
                w[key + ".weight"] = P.pack_conv3x3(ww, dev)
            w[key + ".bias"] = P.pack_bias(b, dev)

        def norm(key):
            w[key + ".weight"], w[key + ".bias"] = P.f32(sd[key + ".weight"], dev), P.f32(sd[key + ".bias"], dev)

        def lin(key):
            w[key + ".weight"] = P.pack_linear(sd[key + ".weight"], dev)
            w[key + ".bias"] = P.pack_bias(sd[key + ".bias"], dev)

        def res(p):
            norm(p + ".norm1"); conv(p + ".conv1"); norm(p + ".norm2"); conv(p + ".conv2", fold_skip=p + ".nin_shortcut")

        def attn(a):
            norm(a + ".norm")
            for nme in ("q", "k", "v", "proj_out"):
                lin(f"{a}.{nme}")

        return conv, norm, lin, res, attn


    # ---- helpers
    def act(self, rows, C, dtype=None):
        return Act(self.pool.get(rows * C * (4 if dtype == torch.float32 else 2)), rows, C, dtype)

    def rel(self, a):
        self.pool.put(a.buf)

    def _gemm(self, label, M, segs, wkey, out, ldo=None, bias=None, N=None, **kw):
        """wkey: name of a packed weight, or a raw device pointer to 16-bit [N][K] rows (then N must be given)."""
        W = self.wt[wkey] if isinstance(wkey, str) else wkey
        if N is None:
            N = W.shape[0]
        if "ksplit" not in kw and not kw.get("rowstat"):
            if getattr(self, "_splitk", None) is None:
                self._splitk = ops.SplitK(self.device if hasattr(self, "device") else self.dev, cap=32)
            kw["ksplit"], kw["workspace"] = self._splitk.pick(M, N, segs)
        self.S.gemm(ops.gemm_params(M, N, segs, W, out.ptr if isinstance(out, Act) else out,
                                    ldo if ldo is not None else out.C, bias=bias, **kw), label)

    def _gn(self, label, x: Act, rows_per_stat, key, silu):
        y = self.act(x.rows, x.C)
        assert ops.gn_partial_floats(x.rows, rows_per_stat, x.C) <= self._gnws.numel()
        self.S.groupnorm(ops.gn_params(x.ptr, x.C, x.C, x.rows, rows_per_stat, self._gnws, self.wt[key + ".weight"],
                                       self.wt[key + ".bias"], 1e-6, silu, y.ptr, y.C), label)
        return y

    def _res(self, p, x: Act, h, w) -> Act:
        T = x.rows
        geom = ops.Geom(OH=h, OW=w, IH=h, IW=w)
        cout = self.wt[p + ".conv1.weight"].shape[0]
        h0 = self._gn(p + ".norm1", x, h * w, p + ".norm1", True)
        h1 = self.act(T, cout)
        self._gemm(p + ".conv1", T, ops.conv3x3_segs([(h0.ptr, h0.C, h0.C)]), p + ".conv1.weight", h1,
                   bias=self.wt[p + ".conv1.bias"], geom=geom)
        self.rel(h0)
        h2 = self._gn(p + ".norm2", h1, h * w, p + ".norm2", True)
        self.rel(h1)
        y = self.act(T, cout)
        segs = ops.conv3x3_segs([(h2.ptr, h2.C, h2.C)])
        if x.C != cout:
            segs += ops.linear_segs([(x.ptr, x.C, x.C)])
            self._gemm(p + ".conv2+nin", T, segs, p + ".conv2.weight", y, bias=self.wt[p + ".conv2.bias"], geom=geom)
        else:
            self._gemm(p + ".conv2", T, segs, p + ".conv2.weight", y, bias=self.wt[p + ".conv2.bias"], geom=geom,
                       residual=x.ptr, ldr=x.C)
        self.rel(h2)
        return y

    def _attn(self, p, x: Act, h, w) -> Act:
        n, hw, C, T = self.n, h * w, x.C, x.rows
        hn = self._gn(p + ".norm", x, hw, p + ".norm", False)
        q, k = self.act(T, C), self.act(T, C)
        lin = ops.linear_segs([(hn.ptr, C, C)])
        self._gemm(p + ".q", T, lin, p + ".q.weight", q, bias=self.wt[p + ".q.bias"])
        self._gemm(p + ".k", T, lin, p + ".k.weight", k, bias=self.wt[p + ".k.bias"])
        hwp = (hw + 7) // 8 * 8
        if hwp != hw:
            raise NotImplementedError("VAE attention needs h*w % 8 == 0")
        ao = self.act(T, C)
        wv = self.wt[p + ".v.weight"]
        # All frames in ONE launch per stage when a 256-row tile never straddles two frames (grouped weights, vmv.h:
        # VmvGemmParams.wgroup_rows — the "weight" operand of Q K^T / P V / V^T is the frame's own K / V^T / h): 4 launches
        # instead of 4 n small ones that fill a quarter of the chip each; frames in groups that keep the fp32 scores <= 1 GiB.
        batched = hw % 256 == 0 and C % 256 == 0 and os.environ.get("VMV_VAE_ATTN_BATCHED", "1") != "0"
        G = max(1, min(n, (1 << 30) // (hw * hwp * 4))) if batched else 1
        vT = self.act(G * C, hwp)                   # V^T of G frames: [g][C][hw]
        sc = self.act(G * hw, hwp, torch.float32)   # scores
        pr = self.act(G * hw, hwp)                  # probabilities (16-bit)
        if batched:
            rep = getattr(self, "_wv_rep", None)
            if rep is None:
                rep = self._wv_rep = {}
            if (p, G) not in rep:
                rep[(p, G)] = wv.repeat(G, 1).contiguous()      # A operand of the grouped V^T GEMM: Wv once per frame
        for f0 in range(0, n, G):
            g = min(G, n - f0)
            off = f0 * hw * C * 2
            if batched:
                # V^T[f][c][j] = sum_k Wv[c][k] hn[f][j][k]   (bias bv is added after P.V: softmax rows sum to 1)
                self._gemm(f"{p}.vT[{f0}+{g}]", g * C, ops.linear_segs([(rep[(p, G)].data_ptr(), C, C)]), hn.ptr + off, vT, ldo=hwp,
                           N=hw, ksplit=0, wgroup_rows=C, wgroup_stride=hw * C)
                self._gemm(f"{p}.qk[{f0}+{g}]", g * hw, ops.linear_segs([(q.ptr + off, C, C)]), k.ptr + off, sc, ldo=hwp,
                           out_fp32=True, N=hw, ksplit=0, wgroup_rows=hw, wgroup_stride=hw * C)
                self.S.softmax(ops.softmax_params(sc.ptr, hwp, pr.ptr, hwp, g * hw, hw, float(C) ** -0.5), f"{p}.softmax[{f0}+{g}]")
                self._gemm(f"{p}.pv[{f0}+{g}]", g * hw, ops.linear_segs([(pr.ptr, hwp, hw)]), vT.ptr, ao.ptr + off, ldo=C,
                           bias=self.wt[p + ".v.bias"], N=C, ksplit=0, wgroup_rows=hw, wgroup_stride=C * hwp)
                continue
            f = f0
            # V^T[c][j] = sum_k Wv[c][k] hn[j][k]   (bias bv is added after P.V: softmax rows sum to 1)
            self._gemm(f"{p}.vT[{f}]", C, ops.linear_segs([(wv.data_ptr(), C, C)]), hn.ptr + off, vT, ldo=hwp, N=hw)
            self._gemm(f"{p}.qk[{f}]", hw, ops.linear_segs([(q.ptr + off, C, C)]), k.ptr + off, sc, ldo=hwp, out_fp32=True,
                       N=hw)
            self.S.softmax(ops.softmax_params(sc.ptr, hwp, pr.ptr, hwp, hw, hw, float(C) ** -0.5), f"{p}.softmax[{f}]")
            self._gemm(f"{p}.pv[{f}]", hw, ops.linear_segs([(pr.ptr, hwp, hw)]), vT.ptr, ao.ptr + off, ldo=C,
                       bias=self.wt[p + ".v.bias"], N=C)
        for a in (hn, q, k, vT, sc, pr):
            self.rel(a)
        y = self.act(T, C)
        self._gemm(p + ".proj_out", T, ops.linear_segs([(ao.ptr, C, C)]), p + ".proj_out.weight", y,
                   bias=self.wt[p + ".proj_out.bias"], residual=x.ptr, ldr=C)
        self.rel(ao)
        return y


class VaeDecoderEngine(_VaeEngine):
    """Recorded plan for ``decode`` of n latent frames of h x w."""

    def _pack(self, sd):
        conv, norm, lin, res, attn = self._packers(sd)
        w = self.wt
        lin("post_quant_conv")
        conv("decoder.conv_in")
        res("decoder.mid.block_1"); res("decoder.mid.block_2")
        attn("decoder.mid.attn_1")
        ch_mult, nrb = list(self.dd["ch_mult"]), self.dd["num_res_blocks"]
        for lvl in range(len(ch_mult)):
            for i in range(nrb + 1):
                res(f"decoder.up.{lvl}.block.{i}")
            if lvl != 0:
                conv(f"decoder.up.{lvl}.upsample.conv")
        norm("decoder.norm_out")
        conv("decoder.conv_out")
        self.out_pad = w["decoder.conv_out.weight"].shape[0]

    def _build(self):
        n, h, w = self.n, self.h, self.w
        dev = self.device
        T = n * h * w
        zc = self.dd["z_channels"]
        self.zpad = (zc + 7) // 8 * 8
        self.out_pad = self.wt["decoder.conv_out.weight"].shape[0]      # (also for an engine that shares another's packed weights)
        self.z_rows = torch.zeros(T, self.zpad, dtype=L.elem(), device=dev)
        self.pq_rows = torch.zeros(T, self.zpad, dtype=L.elem(), device=dev)     # cols >= zc stay zero
        self._gemm("post_quant", T, ops.linear_segs([(self.z_rows.data_ptr(), self.zpad, self.zpad)]),
                   "post_quant_conv.weight", self.pq_rows.data_ptr(), ldo=self.zpad, bias=self.wt["post_quant_conv.bias"])
        c_in = self.wt["decoder.conv_in.weight"].shape[0]
        x = self.act(T, c_in)
        self._gemm("conv_in", T, ops.conv3x3_segs([(self.pq_rows.data_ptr(), self.zpad, self.zpad)]),
                   "decoder.conv_in.weight", x, bias=self.wt["decoder.conv_in.bias"], geom=ops.Geom(OH=h, OW=w, IH=h, IW=w))

        def step(fn, *a):
            nonlocal x
            y = fn(*a)
            self.rel(x)
            x = y

        step(lambda: self._res("decoder.mid.block_1", x, h, w))
        step(lambda: self._attn("decoder.mid.attn_1", x, h, w))
        step(lambda: self._res("decoder.mid.block_2", x, h, w))
        ch_mult, nrb = list(self.dd["ch_mult"]), self.dd["num_res_blocks"]
        for lvl in reversed(range(len(ch_mult))):
            for i in range(nrb + 1):
                step(lambda: self._res(f"decoder.up.{lvl}.block.{i}", x, h, w))
            if lvl != 0:
                p = f"decoder.up.{lvl}.upsample.conv"
                y = self.act(n * 4 * h * w, x.C)
                self._gemm(p, y.rows, ops.conv3x3_segs([(x.ptr, x.C, x.C)]), p + ".weight", y, bias=self.wt[p + ".bias"],
                           geom=ops.Geom(OH=2 * h, OW=2 * w, IH=h, IW=w, ups=1))
                self.rel(x)
                x = y
                h, w = 2 * h, 2 * w
        hn = self._gn("norm_out", x, h * w, "decoder.norm_out", True)
        self.rel(x)
        self.OH, self.OW = h, w
        self.img_rows = torch.zeros(n * h * w, self.out_pad, dtype=torch.float32, device=dev)
        self._gemm("conv_out", n * h * w, ops.conv3x3_segs([(hn.ptr, hn.C, hn.C)]), "decoder.conv_out.weight",
                   self.img_rows.data_ptr(), ldo=self.out_pad, bias=self.wt["decoder.conv_out.bias"],
                   geom=ops.Geom(OH=h, OW=w, IH=h, IW=w), out_fp32=True)
        self.rel(hn)

    def decode(self, z: torch.Tensor) -> torch.Tensor:
        """z [n, zc, h, w] fp32 on device -> [n, out_ch, 8h, 8w] fp32."""
        n, zc, h, w = z.shape
        ops.latent_to_rows(z.reshape(n, zc, 1, h, w).contiguous(), self.z_rows, self.zpad, 1)
        self.S.run()
        out = torch.empty(n, self.out_pad, self.OH, self.OW, dtype=torch.float32, device=self.device)
        ops.rows_to_nchw(self.img_rows, self.out_pad, out)
        return out[:, : self.dd["out_ch"]].contiguous()


class VaeEncoderEngine(_VaeEngine):
    """Recorded plan for ``encode`` of n images of h x w pixels (h, w multiples of 8): conv_in -> 4 levels x 2 ResnetBlocks
    with pad-(0,1,0,1) stride-2 down-sampling convs -> mid (Res, Attn, Res) -> GN + swish -> conv_out -> quant_conv;
    leaves the posterior moments as fp32 rows [n*(h/8)*(w/8), 2*zc]  (Encoder.forward, autoencoder.py:548-576)."""

    def _pack(self, sd):
        conv, norm, lin, res, attn = self._packers(sd)
        conv("encoder.conv_in")
        ch_mult, nrb = list(self.dd["ch_mult"]), self.dd["num_res_blocks"]
        for lvl in range(len(ch_mult)):
            for i in range(nrb):
                res(f"encoder.down.{lvl}.block.{i}")
            if lvl != len(ch_mult) - 1:
                conv(f"encoder.down.{lvl}.downsample.conv")
        res("encoder.mid.block_1"); attn("encoder.mid.attn_1"); res("encoder.mid.block_2")
        norm("encoder.norm_out")
        conv("encoder.conv_out")
        lin("quant_conv")

    def _build(self):
        n, h, w, dev = self.n, self.h, self.w, self.device
        cin = self.dd["in_channels"]
        self.cpad = (cin + 7) // 8 * 8
        self.x_rows = torch.zeros(n * h * w, self.cpad, dtype=L.elem(), device=dev)
        c0 = self.wt["encoder.conv_in.weight"].shape[0]
        x = self.act(n * h * w, c0)
        self._gemm("conv_in", n * h * w, ops.conv3x3_segs([(self.x_rows.data_ptr(), self.cpad, self.cpad)]),
                   "encoder.conv_in.weight", x, bias=self.wt["encoder.conv_in.bias"], geom=ops.Geom(OH=h, OW=w, IH=h, IW=w))
        ch_mult, nrb = list(self.dd["ch_mult"]), self.dd["num_res_blocks"]
        for lvl in range(len(ch_mult)):
            for i in range(nrb):
                y = self._res(f"encoder.down.{lvl}.block.{i}", x, h, w)
                self.rel(x)
                x = y
            if lvl != len(ch_mult) - 1:
                p = f"encoder.down.{lvl}.downsample.conv"
                oh, ow = h // 2, w // 2
                y = self.act(n * oh * ow, x.C)
                self._gemm(p, y.rows, ops.conv3x3_segs([(x.ptr, x.C, x.C)], shift=1), p + ".weight", y,
                           bias=self.wt[p + ".bias"], geom=ops.Geom(OH=oh, OW=ow, IH=h, IW=w, stride=2))
                self.rel(x)
                x, h, w = y, oh, ow
        for fn, p in ((self._res, "encoder.mid.block_1"), (self._attn, "encoder.mid.attn_1"), (self._res, "encoder.mid.block_2")):
            y = fn(p, x, h, w)
            self.rel(x)
            x = y
        hn = self._gn("norm_out", x, h * w, "encoder.norm_out", True)
        self.rel(x)
        T = n * h * w
        co = self.act(T, self.wt["encoder.conv_out.weight"].shape[0])
        self._gemm("conv_out", T, ops.conv3x3_segs([(hn.ptr, hn.C, hn.C)]), "encoder.conv_out.weight", co,
                   bias=self.wt["encoder.conv_out.bias"], geom=ops.Geom(OH=h, OW=w, IH=h, IW=w))
        self.rel(hn)
        self.mc = self.wt["quant_conv.weight"].shape[0]
        self.moment_rows = torch.zeros(T, self.mc, dtype=torch.float32, device=dev)
        self._gemm("quant_conv", T, ops.linear_segs([(co.ptr, co.C, co.C)]), "quant_conv.weight",
                   self.moment_rows.data_ptr(), ldo=self.mc, bias=self.wt["quant_conv.bias"], out_fp32=True)
        self.LH, self.LW = h, w

    def encode(self, x: torch.Tensor) -> torch.Tensor:
        """x [n, 3, H, W] fp32 on device -> moments rows fp32 [n*LH*LW, 2*zc] (valid until the next call)."""
        n, c, h, w = x.shape
        ops.latent_to_rows_keep(x.reshape(n, c, 1, h, w).contiguous(), self.x_rows, self.cpad, 1)
        self.S.run()
        return self.moment_rows


class DiagonalGaussianDistribution(object):
    """Posterior handle returned by ``AutoencoderKL.encode`` (autoencoder.py:213-226): holds the moments produced by the
    HIP encoder; ``sample()`` draws the noise on the HOST RNG exactly like the reference (``torch.randn(shape).to(device)``)
    and evaluates mean + std * noise in ``vmv_posterior_sample``."""

    def __init__(self, moment_rows, n, zc, h, w, device):
        self.moment_rows, self.n, self.zc, self.h, self.w, self.device = moment_rows.clone(), n, zc, h, w, device

    @property
    def parameters(self):
        out = torch.empty(self.n, 2 * self.zc, self.h, self.w, dtype=torch.float32, device=self.device)
        ops.rows_to_nchw(self.moment_rows, 2 * self.zc, out)
        return out

    def sample(self, scale=1.0, parts=1, of=None, rng_part=None):
        """parts > 1: the noise is drawn in that many consecutive host-RNG calls of n / parts images each — what the same number
        of separate encode calls would have drawn (the batched LGM branch encodes both CFG branches at once).
        of = (total, first): the n / parts images of every part are images [first, first + n / parts) of a batch of `total` — the
        noise of the WHOLE batch is drawn (same host-RNG consumption and the same numbers as the unsharded call) and this slice of it
        used: a frame-parallel rank that encodes only its own views of the LGM branch gets the unsharded run's noise for them.
        rng_part = (i, k): this call stands for the i-th of k consecutive encode calls of the unsharded run (CFG-parallel: a branch
        group encodes ONE branch where the unsharded run encodes cond, then uncond): all k draws are consumed, the i-th is used."""
        per = self.n // parts if (parts > 1 and self.n % parts == 0) else self.n
        nparts = self.n // per
        if rng_part is not None:
            if nparts != 1:
                raise ValueError("rng_part stands for whole encode calls: parts must be 1")
            i, k = int(rng_part[0]), int(rng_part[1])
            if not 0 <= i < k:
                raise ValueError("rng_part index outside its count")
            total, first = (int(of[0]), int(of[1])) if of is not None else (self.n, 0)
            if not (0 <= first and first + self.n <= total):
                raise ValueError("posterior sample slice outside its batch")
            draws = [torch.randn(total, self.zc, self.h, self.w) for _ in range(k)]
            noise = self._to_device(draws[i][first:first + self.n].contiguous())
        elif of is not None:
            total, first = int(of[0]), int(of[1])
            if not (0 <= first and first + per <= total):
                raise ValueError("posterior sample slice outside its batch")
            noise = self._to_device(torch.cat([torch.randn(total, self.zc, self.h, self.w)[first:first + per] for _ in range(nparts)]))
        elif nparts > 1:
            noise = self._to_device(torch.cat([torch.randn(per, self.zc, self.h, self.w) for _ in range(nparts)]))
        else:
            noise = self._to_device(torch.randn(self.n, self.zc, self.h, self.w))
        z = torch.empty(self.n, self.zc, self.h, self.w, dtype=torch.float32, device=self.device)
        ops.posterior_sample(self.moment_rows, 2 * self.zc, noise.contiguous(), z, scale)
        return z

    _pinned = {}      # (shape, device) -> (pinned host buffer, event recorded after the last upload from it)

    def _to_device(self, host_noise):
        """Host-RNG noise -> device WITHOUT blocking the host on the stream: `.to(device)` of a pageable tensor waits for everything
        queued before it — here the whole encoder — and the launch that needs the noise then leaves the host late (round 6: 0.5-14 ms
        of idle GPU per LGM-refined step in the kernel trace, tools/experiments/lgm_gaps.py).  The draw goes through a pinned buffer
        and an asynchronous copy; the buffer is reused once the event after its last upload has passed."""
        if not str(self.device).startswith("cuda"):
            return host_noise.to(device=self.device)
        key = (tuple(host_noise.shape), str(self.device))
        ent = DiagonalGaussianDistribution._pinned.get(key)
        if ent is None:
            ent = [torch.empty(host_noise.shape, dtype=torch.float32).pin_memory(), None]
            DiagonalGaussianDistribution._pinned[key] = ent
        if ent[1] is not None:
            ent[1].synchronize()
        ent[0].copy_(host_noise)
        dev = ent[0].to(device=self.device, non_blocking=True)
        ent[1] = torch.cuda.Event()
        ent[1].record(torch.cuda.current_stream(self.device))
        return dev


@AUTO_ENCODER.register_class()
class AutoencoderKL(nn.Module):
    def __init__(self, ddconfig, embed_dim, pretrained=None, ignore_keys=[], image_key="image", colorize_nlabels=None,
                 monitor=None, ema_decay=None, learn_logvar=False, use_vid_decoder=False, **kwargs):
        super().__init__()
        assert ddconfig["double_z"]
        if list(ddconfig.get("attn_resolutions", [])):
            raise NotImplementedError("attn_resolutions != [] is not a VideoMV configuration")
        self.ddconfig, self.embed_dim = dict(ddconfig), embed_dim
        self.learn_logvar, self.image_key = learn_logvar, image_key
        for key, shape in vae_param_shapes(self.ddconfig, embed_dim).items():
            if len(shape) == 1:
                v = torch.zeros(shape) if key.endswith(".bias") else torch.ones(shape)
            else:
                fan = 1
                for d in shape[1:]:
                    fan *= d
                v = torch.empty(shape).normal_(0.0, 1.0 / math.sqrt(fan))
            head, _, rest = key.partition(".")
            child = self._modules.get(head)
            if child is None:
                child = _Holder()
                self.add_module(head, child)
            child.add(rest, nn.Parameter(v, requires_grad=False))
        self._engines = {}
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._engines.clear())
        if pretrained is not None:
            self.init_from_ckpt(pretrained, ignore_keys=ignore_keys)

    frame_independent = True      # decode / encode treat every frame on its own: callers may batch any number of frames

    def init_from_ckpt(self, path, ignore_keys=list()):
        sd = torch.load(path, map_location="cpu")["state_dict"]
        new = collections.OrderedDict()
        for k, v in sd.items():
            if k.find('first_stage_model') >= 0:
                new[k.split('first_stage_model.')[-1]] = v
        self.load_state_dict(new, strict=True)

    def _packed_of(self, kind, device):
        """The packed weights of an existing engine of this kind on this device (engines are dropped on load_state_dict)."""
        for eng in self._engines.values():
            if type(eng) is kind and str(eng.device) == str(device):
                return eng.wt
        return None

    @torch.no_grad()
    def decode(self, z, **kwargs):
        n, zc, h, w = z.shape
        key = (n, h, w, str(z.device))
        eng = self._engines.get(key)
        if eng is None:
            sd = {k: v.detach() for k, v in self.state_dict().items()
                  if k.startswith("decoder.") or k.startswith("post_quant_conv.")}
            eng = VaeDecoderEngine(self.ddconfig, sd, n, h, w, z.device, packed=self._packed_of(VaeDecoderEngine, z.device))
            self._engines[key] = eng
        from .diffusion_ddim import _check_finite
        _check_finite(z, "the latent handed to decode", is_input=True)
        out = eng.decode(z.float())
        _check_finite(out, "the decoded frames")
        return out

    @torch.no_grad()
    def encode(self, x):
        """-> posterior (``.parameters`` [n, 2*zc, h/8, w/8], ``.sample()``); quant_conv(Encoder(x)) on HIP (:76-80)."""
        n, c, h, w = x.shape
        if (h % 8) or (w % 8):
            raise ValueError("image sides must be multiples of 8")
        key = ("enc", n, h, w, str(x.device))
        eng = self._engines.get(key)
        if eng is None:
            sd = {k: v.detach() for k, v in self.state_dict().items() if k.startswith("encoder.") or k.startswith("quant_conv.")}
            eng = VaeEncoderEngine(self.ddconfig, sd, n, h, w, x.device, packed=self._packed_of(VaeEncoderEngine, x.device))
            self._engines[key] = eng
        from .diffusion_ddim import _check_finite
        _check_finite(x, "the image handed to encode", is_input=True)
        rows = eng.encode(x.float())
        return DiagonalGaussianDistribution(rows, n, self.ddconfig["z_channels"], eng.LH, eng.LW, x.device)

    @torch.no_grad()
    def encode_firsr_stage(self, x, scale_factor=1.0, parts=1, of=None, rng_part=None):
        """scale_factor * posterior.sample()  (autoencoder.py:86-91; the typo is the reference's public name)."""
        return self.encode(x).sample(scale=scale_factor, parts=parts, of=of, rng_part=rng_part)

    def forward(self, input, sample_posterior=True):
        raise NotImplementedError("training-time autoencoding is out of scope (inference hot path only)")

"""LGM branch of VideoMV on the gfx950 kernels: the multi-view U-Net that turns 4 decoded views (+ Pluecker rays) into
pixel-aligned 3-D Gaussians (reference ``core/unet.py``, ``core/models.py:87-113``, registered inside the video UNet as
``self.lgm_big``, ``tools/modules/unet/unet_t2v.py:125-129``).

``LgmEngine`` records the forward as a plan of C-ABI launches over channels-last 16-bit rows ``[V*H*W, C]`` like the
other engines: ResnetBlock = GN+SiLU -> conv3x3 -> GN+SiLU -> conv3x3 (+1x1 shortcut folded into the K loop), with the
block's ``(x + res) * skip_scale`` folded into pre-scaled weights and ``res_scale``; the decoder's ``torch.cat`` is a
segment list; down = stride-2 conv, up = nearest-x2 folded into the conv's gather.  MVAttention attends over the
V*H*W tokens of a sample with 16 heads: head_dim 64 (1024-channel levels) runs on the flash kernel, head_dim 32 (512
channels, 4096 tokens) as QK^T GEMM -> row softmax -> PV GEMM per head (K heads gathered head-major by
``vmv_permute_copy``, V^T produced directly by a GEMM with the roles of weights and activations swapped).
The Gaussian activations run in ``vmv_gaussian_activation``; nothing here computes in PyTorch.
"""
import dataclasses
import os
import math
from typing import Dict, Tuple

import torch

from . import _lib as L
from . import ops
from . import packing as P
from .unet_engine import Pool, Act


@dataclasses.dataclass
class LgmOptions:
    """The fields of ``core/options.py`` (``config_defaults['big']``) that inference reads."""
    in_channels: int = 9
    out_channels: int = 14
    down_channels: Tuple[int, ...] = (64, 128, 256, 512, 1024, 1024)
    down_attention: Tuple[bool, ...] = (False, False, False, True, True, True)
    mid_attention: bool = True
    up_channels: Tuple[int, ...] = (1024, 1024, 512, 256, 128)
    up_attention: Tuple[bool, ...] = (True, True, True, False, False)
    layers_per_block: int = 2
    num_heads: int = 16
    num_frames: int = 4
    skip_scale: float = math.sqrt(0.5)
    input_size: int = 256
    splat_size: int = 128
    output_size: int = 512
    fovy: float = 39.6
    znear: float = 0.5
    zfar: float = 2.5


def _res_shapes(p, cin, cout):
    s = [(f"{p}.norm1.weight", (cin,)), (f"{p}.norm1.bias", (cin,)),
         (f"{p}.conv1.weight", (cout, cin, 3, 3)), (f"{p}.conv1.bias", (cout,)),
         (f"{p}.norm2.weight", (cout,)), (f"{p}.norm2.bias", (cout,)),
         (f"{p}.conv2.weight", (cout, cout, 3, 3)), (f"{p}.conv2.bias", (cout,))]
    if cin != cout:
        s += [(f"{p}.shortcut.weight", (cout, cin, 1, 1)), (f"{p}.shortcut.bias", (cout,))]
    return s


def _attn_shapes(p, c):
    return [(f"{p}.norm.weight", (c,)), (f"{p}.norm.bias", (c,)), (f"{p}.attn.qkv.weight", (3 * c, c)),
            (f"{p}.attn.proj.weight", (c, c)), (f"{p}.attn.proj.bias", (c,))]


def lgm_param_shapes(o: LgmOptions) -> Dict[str, tuple]:
    """state-dict manifest of ``LGM`` (``unet.*`` + ``conv.*``) in registration order; equals
    tests/golden/manifest_lgm_big.json for the default options."""
    s = [("conv_in.weight", (o.down_channels[0], o.in_channels, 3, 3)), ("conv_in.bias", (o.down_channels[0],))]
    cout, nd, nu = o.down_channels[0], len(o.down_channels), len(o.up_channels)
    for i in range(nd):
        cin, cout = cout, o.down_channels[i]
        for j in range(o.layers_per_block):
            s += _res_shapes(f"down_blocks.{i}.nets.{j}", cin if j == 0 else cout, cout)
        if o.down_attention[i]:
            for j in range(o.layers_per_block):
                s += _attn_shapes(f"down_blocks.{i}.attns.{j}", cout)
        if i != nd - 1:
            s += [(f"down_blocks.{i}.downsample.weight", (cout, cout, 3, 3)), (f"down_blocks.{i}.downsample.bias", (cout,))]
    cm = o.down_channels[-1]
    s += _res_shapes("mid_block.nets.0", cm, cm) + _res_shapes("mid_block.nets.1", cm, cm)
    if o.mid_attention:
        s += _attn_shapes("mid_block.attns.0", cm)
    cout = o.up_channels[0]
    for i in range(nu):
        cin, cout = cout, o.up_channels[i]
        cskip = o.down_channels[max(-2 - i, -nd)]
        nl = o.layers_per_block + 1
        for j in range(nl):
            s += _res_shapes(f"up_blocks.{i}.nets.{j}", (cin if j == 0 else cout) + (cskip if j == nl - 1 else cout), cout)
        if o.up_attention[i]:
            for j in range(nl):
                s += _attn_shapes(f"up_blocks.{i}.attns.{j}", cout)
        if i != nu - 1:
            s += [(f"up_blocks.{i}.upsample.weight", (cout, cout, 3, 3)), (f"up_blocks.{i}.upsample.bias", (cout,))]
    s += [("norm_out.weight", (o.up_channels[-1],)), ("norm_out.bias", (o.up_channels[-1],)),
          ("conv_out.weight", (o.out_channels, o.up_channels[-1], 3, 3)), ("conv_out.bias", (o.out_channels,))]
    d = {("unet." + k): v for k, v in s}
    d["conv.weight"], d["conv.bias"] = (14, 14, 1, 1), (14,)
    return d


class LgmEngine:
    """Plan for ``LGM.forward_gaussians`` of ``batch`` samples: images [batch * V, 9, H, W] -> gaussians fp32
    [batch * V*S*S, 14] (sample-major).  Everything but the multi-view attention is per image; the attention runs over the
    V*h*w tokens of each sample.  ``packed``: the .wt of another engine of the same model and device (one weight copy)."""

    def __init__(self, opt: LgmOptions, sd: Dict[str, torch.Tensor], H: int, W: int, device, taps=None, batch: int = 1, packed=None):
        self.o, self.B, self.H, self.W, self.device = opt, int(batch), H, W, device
        self.V = opt.num_frames * self.B             # images in the plan (rows are image-major)
        self.pool = Pool(device)
        self.S = ops.Stream(record=True)
        self.S.tuner = ops.make_tuner(self)      # measured per-shape (tile, split-K) choices: videomv_amd/tuned_gemm.json
        self._gnws = torch.empty(4 << 20, dtype=torch.float32, device=device)
        self._keep = []
        self.taps = taps
        if packed is not None:
            self.wt = packed
        else:
            self.wt: Dict[str, torch.Tensor] = {}
            self._pack(sd)
        self._build()

    # ------------------------------------------------------------------ weights
    def _pack(self, sd):
        dev, w, k = self.device, self.wt, self.o.skip_scale
        u = lambda key: sd["unet." + key]

        def norm(p):
            w[p + ".weight"], w[p + ".bias"] = P.f32(u(p + ".weight"), dev), P.f32(u(p + ".bias"), dev)

        def conv(p):
            w[p + ".weight"], w[p + ".bias"] = P.pack_conv3x3(u(p + ".weight"), dev), P.pack_bias(u(p + ".bias"), dev)

        def res(p):
            norm(p + ".norm1"); conv(p + ".conv1"); norm(p + ".norm2")
            w2 = u(p + ".conv2.weight")
            b2 = u(p + ".conv2.bias").float()
            w2p = w2.permute(0, 2, 3, 1).reshape(w2.shape[0], -1)
            if ("unet." + p + ".shortcut.weight") in sd:          # 1x1 shortcut joins conv2's K loop
                w2p = torch.cat([w2p, u(p + ".shortcut.weight").reshape(w2.shape[0], -1)], dim=1)
                b2 = b2 + u(p + ".shortcut.bias").float()
            w[p + ".conv2.weight"] = P.pack_linear(w2p * k, dev)   # (conv2 + shortcut) * skip_scale
            w[p + ".conv2.bias"] = P.pack_bias(b2 * k, dev)

        def attn(p):
            norm(p + ".norm")
            qkv = u(p + ".attn.qkv.weight")
            C = qkv.shape[1]
            w[p + ".qkv"] = P.pack_linear(qkv, dev)
            w[p + ".wv"] = P.pack_linear(qkv[2 * C:], dev)          # rows of W_v: the A operand of the V^T GEMM (head_dim 32)
            w[p + ".proj.weight"] = P.pack_linear(u(p + ".attn.proj.weight") * k, dev)
            w[p + ".proj.bias"] = P.pack_bias(u(p + ".attn.proj.bias").float() * k, dev)

        o = self.o
        nd, nu = len(o.down_channels), len(o.up_channels)
        conv("conv_in")
        for i in range(nd):
            for j in range(o.layers_per_block):
                res(f"down_blocks.{i}.nets.{j}")
                if o.down_attention[i]:
                    attn(f"down_blocks.{i}.attns.{j}")
            if i != nd - 1:
                conv(f"down_blocks.{i}.downsample")
        res("mid_block.nets.0"); res("mid_block.nets.1")
        if o.mid_attention:
            attn("mid_block.attns.0")
        for i in range(nu):
            for j in range(o.layers_per_block + 1):
                res(f"up_blocks.{i}.nets.{j}")
                if o.up_attention[i]:
                    attn(f"up_blocks.{i}.attns.{j}")
            if i != nu - 1:
                conv(f"up_blocks.{i}.upsample")
        norm("norm_out"); conv("conv_out")
        w["final.weight"] = P.pack_linear(sd["conv.weight"].reshape(14, 14), dev)      # LGM.conv (1x1), K padded 14 -> 16
        w["final.bias"] = P.pack_bias(sd["conv.bias"], dev)

    # ------------------------------------------------------------------ helpers
    def act(self, rows, C, dtype=None):
        return Act(self.pool.get(rows * C * (4 if dtype == torch.float32 else 2)), rows, C, dtype)

    def rel(self, a):
        if self.taps is None:
            self.pool.put(a.buf)

    def _gemm(self, label, M, segs, W, out, ldo=None, bias=None, N=None, **kw):
        Wt = self.wt[W] if isinstance(W, str) else W
        if N is None:
            N = Wt.shape[0]
        if "ksplit" not in kw and not kw.get("rowstat"):
            if getattr(self, "_splitk", None) is None:
                self._splitk = ops.SplitK(self.device if hasattr(self, "device") else self.dev, cap=32)
            kw["ksplit"], kw["workspace"] = self._splitk.pick(M, N, segs)
        self.S.gemm(ops.gemm_params(M, N, segs, Wt, out.ptr if isinstance(out, Act) else out,
                                    ldo if ldo is not None else out.C, bias=bias, **kw), label)

    def _gn(self, label, srcs, hw, key, silu) -> Act:
        rows = srcs[0].rows
        C0 = srcs[0].C
        C1 = srcs[1].C if len(srcs) > 1 else 0
        y = self.act(rows, C0 + C1)
        assert ops.gn_partial_floats(rows, hw, C0 + C1) <= self._gnws.numel()
        self.S.groupnorm(ops.gn_params(srcs[0].ptr, C0, C0, rows, hw, self._gnws, self.wt[key + ".weight"],
                                       self.wt[key + ".bias"], 1e-5, silu, y.ptr, C0 + C1,
                                       x1=srcs[1].ptr if C1 else None, ld1=C1, C1=C1), label)
        return y

    def _res(self, p, srcs, h, w) -> Act:
        """srcs = [x] or [x, skip] (channel concat, never materialised)."""
        T = srcs[0].rows
        geom = ops.Geom(OH=h, OW=w, IH=h, IW=w)
        cout = self.wt[p + ".conv1.weight"].shape[0]
        h0 = self._gn(p + ".norm1", srcs, h * w, p + ".norm1", True)
        h1 = self.act(T, cout)
        self._gemm(p + ".conv1", T, ops.conv3x3_segs([(h0.ptr, h0.C, h0.C)]), p + ".conv1.weight", h1,
                   bias=self.wt[p + ".conv1.bias"], geom=geom)
        self.rel(h0)
        h2 = self._gn(p + ".norm2", [h1], h * w, p + ".norm2", True)
        self.rel(h1)
        y = self.act(T, cout)
        segs = ops.conv3x3_segs([(h2.ptr, h2.C, h2.C)])
        cin = sum(s.C for s in srcs)
        if cin != cout:
            segs += ops.linear_segs([(s.ptr, s.C, s.C) for s in srcs])
            self._gemm(p + ".conv2+shortcut", T, segs, p + ".conv2.weight", y, bias=self.wt[p + ".conv2.bias"], geom=geom)
        else:           # identity shortcut: (conv2(h) + x) * k = k*conv2(h) + k*x
            self._gemm(p + ".conv2", T, segs, p + ".conv2.weight", y, bias=self.wt[p + ".conv2.bias"], geom=geom,
                       residual=srcs[0].ptr, ldr=srcs[0].C, res_scale=self.o.skip_scale)
        self.rel(h2)
        return y

    def _attn(self, p, x: Act, h, w) -> Act:
        o = self.o
        C, T = x.C, x.rows // self.B               # T = V*h*w tokens of one sample
        heads = o.num_heads
        hd = C // heads
        hn = self._gn(p + ".norm", [x], h * w, p + ".norm", False)
        qkv = self.act(x.rows, 3 * C)
        self._gemm(p + ".qkv", x.rows, ops.linear_segs([(hn.ptr, C, C)]), p + ".qkv", qkv)
        ao = self.act(x.rows, C)
        if hd == 64 or (hd == 32 and os.environ.get("VMV_ATTN_D32", "1") != "0"):
            self.rel(hn)                            # flash kernel (head_dim 64, and 32: the 'big' model's 512 / 16 heads)
            m = lambda: ops.seq_map(T * 3 * C, 0, 3 * C, inner=1)
            self.S.attention(ops.attn_params(qkv.ptr, qkv.ptr + 2 * C, qkv.ptr + 4 * C, ao.ptr, m(), m(), m(),
                                             ops.seq_map(T * C, 0, C, inner=1), self.B, heads, T, T, hd ** -0.5, head_dim=hd),
                             p + ".attn")
        else:
            if self.B != 1:
                raise NotImplementedError("batched LGM plans need head_dim 64 or 32 (the flash kernel)")
            if hd % 8 or T % 8:
                raise NotImplementedError("LGM attention: head_dim and token count must be multiples of 8")
            # V^T[c][j] = sum_k Wv[c][k] hn[j][k]  -> [C][T]; K gathered head-major [heads][T][hd]
            vT = self.act(C, T)
            self._gemm(p + ".vT", C, ops.linear_segs([(self.wt[p + ".wv"].data_ptr(), C, C)]), hn.ptr, vT, ldo=T, N=T)
            self.rel(hn)
            kh = self.act(heads * T, hd)
            self.S.copy(ops.copy_params(qkv.ptr + 2 * C, kh.ptr, heads, T, 1, hd // 8, hd // 8, 3 * C // 8), p + ".k_heads")
            sc = self.act(T, T, torch.float32)
            pr = self.act(T, T)
            for hh in range(heads):
                self._gemm(f"{p}.qk[{hh}]", T, ops.linear_segs([(qkv.ptr + 2 * hh * hd, 3 * C, hd)]), kh.ptr + 2 * hh * T * hd,
                           sc, ldo=T, out_fp32=True, N=T)
                self.S.softmax(ops.softmax_params(sc.ptr, T, pr.ptr, T, T, T, hd ** -0.5), f"{p}.softmax[{hh}]")
                self._gemm(f"{p}.pv[{hh}]", T, ops.linear_segs([(pr.ptr, T, T)]), vT.ptr + 2 * hh * hd * T,
                           ao.ptr + 2 * hh * hd, ldo=C, N=hd)
            for a in (vT, kh, sc, pr):
                self.rel(a)
        self.rel(qkv)
        y = self.act(x.rows, C)
        self._gemm(p + ".proj", x.rows, ops.linear_segs([(ao.ptr, C, C)]), p + ".proj.weight", y, bias=self.wt[p + ".proj.bias"],
                   residual=x.ptr, ldr=C, res_scale=o.skip_scale)
        self.rel(ao)
        return y

    # ------------------------------------------------------------------ plan
    def _build(self):
        o, V, dev = self.o, self.V, self.device
        h, w = self.H, self.W
        nd, nu = len(o.down_channels), len(o.up_channels)
        self.cin_pad = (o.in_channels + 7) // 8 * 8
        self.x_rows = torch.zeros(V * h * w, self.cin_pad, dtype=L.elem(), device=dev)
        x = self.act(V * h * w, o.down_channels[0])
        self._gemm("conv_in", x.rows, ops.conv3x3_segs([(self.x_rows.data_ptr(), self.cin_pad, self.cin_pad)]),
                   "conv_in.weight", x, bias=self.wt["conv_in.bias"], geom=ops.Geom(OH=h, OW=w, IH=h, IW=w))
        xss = [(x, h, w)]                              # skip stack: activations stay alive until the decoder pops them
        for i in range(nd):
            for j in range(o.layers_per_block):
                y = self._res(f"down_blocks.{i}.nets.{j}", [x], h, w)
                if o.down_attention[i]:
                    y2 = self._attn(f"down_blocks.{i}.attns.{j}", y, h, w)
                    self.rel(y)
                    y = y2
                x = y
                xss.append((x, h, w))
            if i != nd - 1:
                oh, ow = (h + 1) // 2, (w + 1) // 2
                p = f"down_blocks.{i}.downsample"
                y = self.act(V * oh * ow, x.C)
                self._gemm(p, y.rows, ops.conv3x3_segs([(x.ptr, x.C, x.C)]), p + ".weight", y, bias=self.wt[p + ".bias"],
                           geom=ops.Geom(OH=oh, OW=ow, IH=h, IW=w, stride=2))
                x, h, w = y, oh, ow
                xss.append((x, h, w))
            if self.taps is not None:
                self.taps[f"down_blocks.{i}"] = (x, h, w)
        y = self._res("mid_block.nets.0", [x], h, w)       # x itself is the top of the skip stack: not released here
        if o.mid_attention:
            y2 = self._attn("mid_block.attns.0", y, h, w)
            self.rel(y)
            y = y2
        x = self._res("mid_block.nets.1", [y], h, w)
        self.rel(y)
        if self.taps is not None:
            self.taps["mid_block"] = (x, h, w)
        for i in range(nu):
            nl = o.layers_per_block + 1
            for j in range(nl):
                skip, sh, sw = xss.pop()
                assert (sh, sw) == (h, w), (sh, sw, h, w)
                y = self._res(f"up_blocks.{i}.nets.{j}", [x, skip], h, w)
                self.rel(x)
                self.rel(skip)
                if o.up_attention[i]:
                    y2 = self._attn(f"up_blocks.{i}.attns.{j}", y, h, w)
                    self.rel(y)
                    y = y2
                x = y
            if i != nu - 1:
                p = f"up_blocks.{i}.upsample"
                y = self.act(V * 4 * h * w, x.C)
                self._gemm(p, y.rows, ops.conv3x3_segs([(x.ptr, x.C, x.C)]), p + ".weight", y, bias=self.wt[p + ".bias"],
                           geom=ops.Geom(OH=2 * h, OW=2 * w, IH=h, IW=w, ups=1))
                self.rel(x)
                x, h, w = y, 2 * h, 2 * w
            if self.taps is not None:
                self.taps[f"up_blocks.{i}"] = (x, h, w)
        for leftover, _, _ in xss:            # the 'big' U-Net is asymmetric (5 decoder levels for 6 encoder levels): the
            self.rel(leftover)                # shallowest skips are simply unused, as in the reference (core/unet.py:306-310)
        hn = self._gn("norm_out", [x], h * w, "norm_out", True)
        self.rel(x)
        T = V * h * w
        self.S_out, self.T_out = h, T
        self._u14 = torch.zeros(T, 16, dtype=L.elem(), device=dev)         # conv_out: 14 channels at row pitch 16 (the K of
        u14 = Act(self._u14.view(torch.uint8).view(-1), T, 16)         # the final 1x1); the packed weight's 2 pad rows give 0
        self._gemm("conv_out", T, ops.conv3x3_segs([(hn.ptr, hn.C, hn.C)]), "conv_out.weight", u14, ldo=16,
                   bias=self.wt["conv_out.bias"], geom=ops.Geom(OH=h, OW=w, IH=h, IW=w))
        self.rel(hn)
        self.raw_rows = torch.zeros(T, 16, dtype=torch.float32, device=dev)
        self._gemm("final_1x1", T, ops.linear_segs([(u14.ptr, 16, 16)]), "final.weight", self.raw_rows.data_ptr(), ldo=16,
                   bias=self.wt["final.bias"], out_fp32=True)
        self.gaussians = torch.zeros(T, 14, dtype=torch.float32, device=dev)
        self._act_ws = torch.zeros(1024, dtype=torch.float32, device=dev)

    # ------------------------------------------------------------------ run
    def forward_gaussians(self, images: torch.Tensor) -> torch.Tensor:
        """images [batch * V, 9, H, W] fp32 on device -> gaussians [batch * V*S*S, 14] fp32 (pos, opacity, scale, rotation, rgb)."""
        V, Cc, H, W = images.shape
        assert (V, H, W) == (self.V, self.H, self.W) and Cc == self.o.in_channels
        ops.latent_to_rows(images.reshape(V, Cc, 1, H, W).contiguous(), self.x_rows, self.cin_pad, 1)
        self.S.run()
        ops.gaussian_activation(self.raw_rows, 16, self.gaussians, self.T_out, self._act_ws)
        return self.gaussians


# --------------------------------------------------------------------------------------------- cameras / rays (host, once per prompt)
def get_rays(pose, h, w, fovy):
    """core/utils.py:10-43 (OpenGL convention): origins and unit directions [h, w, 3] of a camera-to-world pose."""
    x, y = torch.meshgrid(torch.arange(w), torch.arange(h), indexing="xy")
    x, y = x.flatten().to(pose.dtype), y.flatten().to(pose.dtype)
    focal = h * 0.5 / math.tan(0.5 * math.radians(fovy))
    dirs = torch.stack([(x - w * 0.5 + 0.5) / focal, -(y - h * 0.5 + 0.5) / focal, -torch.ones_like(x)], dim=-1)
    rays_d = dirs @ pose[:3, :3].transpose(0, 1)
    rays_o = pose[:3, 3].unsqueeze(0).expand_as(rays_d)
    rays_d = rays_d / torch.sqrt(torch.clamp((rays_d * rays_d).sum(-1, keepdim=True), min=1e-20))
    return rays_o.reshape(h, w, 3), rays_d.reshape(h, w, 3)


def prepare_gs_data(camera_data: torch.Tensor, opt: LgmOptions = None) -> dict:
    """camera_data [1, T, 16] (the UNet's camera condition) -> gs_data = dict(input [1,T,6,S,S] Pluecker rays, cam_view,
    cam_view_proj [1,T,4,4], cam_pos [1,T,3]) exactly as the entrance builds it
    (tools/inferences/inference_text2video_entrance.py:198-235), conventions included."""
    opt = opt or LgmOptions()
    T = camera_data.shape[1]
    cam = camera_data.detach().cpu().clone().reshape(T, 4, 4).contiguous().float()
    cam[:, 1] *= -1
    cam[:, [1, 2]] = cam[:, [2, 1]]
    cam[:, :3, 1:3] *= -1
    dist = float(torch.sqrt(cam[0, 0, 3] ** 2 + cam[0, 1, 3] ** 2 + cam[0, 2, 3] ** 2))
    transform = torch.tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, dist], [0, 0, 0, 1]], dtype=torch.float32) @ torch.inverse(cam[0])
    poses = transform.unsqueeze(0) @ cam
    rays = []
    for i in range(T):
        o, d = get_rays(poses[i], opt.input_size, opt.input_size, opt.fovy)
        rays.append(torch.cat([torch.cross(o, d, dim=-1), d], dim=-1))
    rays = torch.stack(rays, dim=0).permute(0, 3, 1, 2).contiguous()
    tan = math.tan(0.5 * math.radians(opt.fovy))
    proj = torch.zeros(4, 4)
    proj[0, 0] = proj[1, 1] = 1 / tan
    proj[2, 2] = (opt.zfar + opt.znear) / (opt.zfar - opt.znear)
    proj[3, 2] = -(opt.zfar * opt.znear) / (opt.zfar - opt.znear)
    proj[2, 3] = 1
    poses = poses.clone()
    poses[:, :3, 1:3] *= -1
    view = torch.inverse(poses).transpose(1, 2)
    return dict(input=rays.unsqueeze(0), cam_view=view.unsqueeze(0), cam_view_proj=(view @ proj).unsqueeze(0),
                cam_pos=(-poses[:, :3, 3]).unsqueeze(0))


class LgmRefiner:
    """The ``autoencoder is not None`` branch of ``UNetSD_T2VBase.forward`` (unet_t2v.py:404-433) for ONE branch of ONE
    sample: predicted x0 of 4 views -> VAE decode -> LGM Gaussians -> 24 renders -> VAE encode -> latent_z."""

    def __init__(self, opt: LgmOptions, lgm_state: Dict[str, torch.Tensor], device, bg_color=0.5):
        from .gs import GaussianRenderer
        self.opt, self.device, self.bg_color = opt, device, float(bg_color)
        self.engine = LgmEngine(opt, lgm_state, opt.input_size, opt.input_size, device)
        self._engine2 = None                  # two samples per plan (both CFG branches at once), same packed weights
        self._dev_cache = {}                  # device copies of gs_data's per-sample constants (_dev_const)
        self._lgm_state = lgm_state
        self.renderer = GaussianRenderer(opt.output_size, opt.fovy, opt.znear, opt.zfar)
        S = opt.input_size
        self.z4 = None
        self.z8 = None
        self.inp = torch.zeros(opt.num_frames, 9, S, S, dtype=torch.float32, device=device)
        self.inp2 = None
        self.last_gaussians = None

    def pair_supported(self) -> bool:
        hd_ok = all((c // self.opt.num_heads) in (32, 64) for c, a in
                    list(zip(self.opt.down_channels, self.opt.down_attention)) + list(zip(self.opt.up_channels, self.opt.up_attention))
                    + [(self.opt.down_channels[-1], self.opt.mid_attention)] if a)
        return hd_ok and os.environ.get("VMV_LGM_BATCHED", "1") != "0"

    @torch.no_grad()
    def _dev_const(self, gs_data, name, make):
        """Device copy of one of gs_data's per-sample constants (rays of the key views, cameras), made once per tensor instead of once
        per refined step: a blocking `.to(device)` of a host tensor drains the stream, and the launches after it leave the host late."""
        src = gs_data["input"] if name == "rays" else gs_data[name]
        if src.is_cuda and name != "rays":
            return src
        key = (name, src.data_ptr(), tuple(src.shape), src._version)
        hit = self._dev_cache.get(name)
        if hit is None or hit[0] != key:
            hit = (key, make(), src)           # (src kept alive: its address is the key)
            self._dev_cache[name] = hit
        return hit[1]

    def latent_z_pair(self, eps_rows, ld, xt, c_recip, c_recipm1, autoencoder, gs_data, scale_factor=0.18215, views=None):
        """latent_z of BOTH CFG branches (rows blocks 0 and 1 of ``eps_rows``) with every stage batched over the two: one VAE
        decode of 8 views, one LGM plan of two samples, 2 x 24 renders, one VAE encode of 48 views.  Same arithmetic per image as
        two ``latent_z`` calls (the posterior noise is drawn in the same two host-RNG calls); the GEMMs see twice the rows, so
        fewer, fuller launches."""
        _, Cc, F_, h, w = xt.shape
        idxs = [0, 6, 12, 18] if F_ == 24 else [i * F_ // 4 for i in range(4)]
        V = len(idxs)
        if self.z8 is None or self.z8.shape[-2:] != (h, w):
            self.z8 = torch.zeros(2 * V, Cc, h, w, dtype=torch.float32, device=self.device)
        for br in range(2):
            ops.lgm_x0_views(eps_rows, ld, br, xt, idxs, c_recip, c_recipm1, 1.0 / scale_factor, self.z8[br * V:(br + 1) * V])
        decoded = autoencoder.decode(self.z8)                                           # [8, 3, S, S]
        S = self.opt.input_size
        if decoded.shape[-1] != S or decoded.shape[-2] != S:
            raise ValueError(f"LGM expects {S}x{S} decoded views (latent {S // 8}x{S // 8}), got {tuple(decoded.shape[-2:])}")
        if self._engine2 is None:
            self._engine2 = LgmEngine(self.opt, self._lgm_state, S, S, self.device, batch=2, packed=self.engine.wt)
            self.inp2 = torch.zeros(2 * self.opt.num_frames, 9, S, S, dtype=torch.float32, device=self.device)
        rays = self._dev_const(gs_data, "rays", lambda: gs_data["input"][0, idxs].to(self.device, torch.float32).contiguous())
        decoded = decoded.contiguous()
        for br in range(2):
            ops.lgm_pack_input(decoded[br * V:(br + 1) * V], rays, self.inp2[br * V:(br + 1) * V])
        gaussians = self._engine2.forward_gaussians(self.inp2).view(2, -1, 14)
        self.last_gaussians = gaussians[0]          # (profiling handle: bench.py times the rasteriser alone on these)
        bg = torch.full((3,), self.bg_color, dtype=torch.float32, device=self.device)
        cv = self._dev_const(gs_data, "cam_view", lambda: gs_data["cam_view"].to(self.device))
        cvp = self._dev_const(gs_data, "cam_view_proj", lambda: gs_data["cam_view_proj"].to(self.device))
        if h != w:
            raise ValueError("the LGM branch renders square views")
        of = None
        if views is not None:      # frame-parallel: x0 / decode / LGM U-Net are the whole sample's (replicated on every rank), the renders
            f0, cnt = int(views[0]), int(views[1])      # and the re-encode only this rank's views [f0, f0 + cnt)
            cv, cvp, of = cv[:, f0:f0 + cnt].contiguous(), cvp[:, f0:f0 + cnt].contiguous(), (cv.shape[1], f0)
        # both branches' views in ONE rasteriser pass (gs.py: B = 2 samples x T views; one host read of the instance total per step)
        T = cv.shape[1]
        both = self.renderer.render(gaussians, cv.expand(2, -1, -1, -1), cvp.expand(2, -1, -1, -1), None, bg_color=bg)["image"]
        small = torch.empty(2 * T, 3, 8 * h, 8 * w, dtype=torch.float32, device=self.device)
        ops.lgm_render_to_vae(both.reshape(2 * T, *both.shape[2:]), small)
        z = autoencoder.encode_firsr_stage(small, scale_factor, parts=2, of=of)         # [2T, C, h, w]
        z = z.reshape(2, 1, T, z.shape[1], z.shape[2], z.shape[3]).permute(0, 1, 3, 2, 4, 5).contiguous()
        return z[0], z[1]

    @torch.no_grad()
    def latent_z(self, eps_rows, ld, branch, xt, c_recip, c_recipm1, autoencoder, gs_data, scale_factor=0.18215, views=None,
                 rng_part=None):
        """x0 = c_recip * xt - c_recipm1 * pred  (eps-prediction: sqrt(1/a), sqrt(1/a - 1); v-prediction: sqrt(a), sqrt(1-a))."""
        _, Cc, F_, h, w = xt.shape
        idxs = [0, 6, 12, 18] if F_ == 24 else [i * F_ // 4 for i in range(4)]          # unet_t2v.py:409 (F = 24)
        if self.z4 is None or self.z4.shape[-2:] != (h, w):
            self.z4 = torch.zeros(4, Cc, h, w, dtype=torch.float32, device=self.device)
        ops.lgm_x0_views(eps_rows, ld, branch, xt, idxs, c_recip, c_recipm1, 1.0 / scale_factor, self.z4)
        decoded = autoencoder.decode(self.z4)                                           # [4, 3, S, S] in [-1, 1]
        S = self.opt.input_size
        if decoded.shape[-1] != S or decoded.shape[-2] != S:
            raise ValueError(f"LGM expects {S}x{S} decoded views (latent {S // 8}x{S // 8}), got {tuple(decoded.shape[-2:])}")
        rays = self._dev_const(gs_data, "rays", lambda: gs_data["input"][0, idxs].to(self.device, torch.float32).contiguous())
        ops.lgm_pack_input(decoded.contiguous(), rays, self.inp)
        gaussians = self.engine.forward_gaussians(self.inp)
        self.last_gaussians = gaussians
        bg = torch.full((3,), self.bg_color, dtype=torch.float32, device=self.device)   # LGM.infer bg_color_factor
        cv = self._dev_const(gs_data, "cam_view", lambda: gs_data["cam_view"].to(self.device))
        cvp, of = self._dev_const(gs_data, "cam_view_proj", lambda: gs_data["cam_view_proj"].to(self.device)), None
        if views is not None:      # (frame-parallel: this rank's views only — see latent_z_pair)
            f0, cnt = int(views[0]), int(views[1])
            cv, cvp, of = cv[:, f0:f0 + cnt].contiguous(), cvp[:, f0:f0 + cnt].contiguous(), (cv.shape[1], f0)
        out = self.renderer.render(gaussians.unsqueeze(0), cv, cvp, None, bg_color=bg)
        images = out["image"][0]                                                        # [T, 3, 2S, 2S]
        T = images.shape[0]
        # F.interpolate(images, (256, 256), mode='nearest') in the reference (unet_t2v.py:425-427) = 8 x the latent size here,
        # whatever the render size (output_size need not be twice it)
        small = torch.empty(T, 3, 8 * h, 8 * w, dtype=torch.float32, device=self.device)
        if h != w:
            raise ValueError("the LGM branch renders square views")
        ops.lgm_render_to_vae(images.contiguous(), small)
        kw = {} if rng_part is None else dict(rng_part=rng_part)      # (a foreign autoencoder need not know the keyword)
        z = autoencoder.encode_firsr_stage(small, scale_factor, of=of, **kw)            # [T, C, h, w]
        return z.reshape(1, T, z.shape[1], z.shape[2], z.shape[3]).permute(0, 2, 1, 3, 4).contiguous()

"""ORACLE (test infrastructure, CPU, fp32) — restatement of the reference video-UNet forward.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product path (``videomv_amd``) never does.  Parity is PINNED: ``tests/golden/*.safetensors``
hold outputs of the imported reference (``oracle/make_golden.py``) and ``tests/test_oracle_golden.py``
checks this file against them to 1e-5.

It is a functional, state-dict driven restatement (channels-first fp32, plain torch ops) of:
  * ``UNetSD_T2VBase.forward``               tools/modules/unet/unet_t2v.py:283-403 (+ ctor :141-265)
  * ``ResBlock._forward``                    tools/modules/unet/util.py:703-730
  * ``TemporalConvBlock_v2.forward``         tools/modules/unet/util.py:1381-1392
  * ``SpatialTransformer.forward``           tools/modules/unet/util.py:354-373
  * ``TemporalTransformer.forward``          tools/modules/unet/util.py:1043-1089
  * ``BasicTransformerBlock.forward``        tools/modules/unet/util.py:536-540
  * ``MemoryEfficientCrossAttention``        tools/modules/unet/util.py:230-268 (exact softmax attention)
  * ``GEGLU`` / ``FeedForward``              tools/modules/unet/util.py:541-576
  * ``Upsample`` / ``Downsample``            tools/modules/unet/util.py:597-607, 754-756
  * ``sinusoidal_embedding``                 tools/modules/unet/util.py:177-189
The parameter names are the reference's state-dict keys (typos are API).
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F


@dataclass
class UNetCfg:
    """Architecture hyper-parameters (python defaults merged with the YAML)."""
    in_dim: int = 4
    dim: int = 320
    context_dim: int = 1024
    out_dim: int = 4
    dim_mult: List[int] = field(default_factory=lambda: [1, 2, 4, 4])
    num_heads: int = 8
    head_dim: int = 64
    num_res_blocks: int = 2
    attn_scales: List[float] = field(default_factory=lambda: [1.0, 0.5, 0.25])
    camera_dim: int = 16
    use_camera_condition: bool = True
    use_fps_condition: bool = False
    dec_context_dim: int = 1024  # hard-coded in the decoder (unet_t2v.py:226)


def block_plan(cfg: UNetCfg):
    """Enumerate the block structure exactly as the constructor builds it (unet_t2v.py:160-258).

    Returns (input_blocks, middle, output_blocks); each block is a list of
    (kind, prefix, meta) with kind in {conv_in, tt, res, st, down, up}.
    """
    dim = cfg.dim
    enc_dims = [dim * u for u in [1] + cfg.dim_mult]
    dec_dims = [dim * u for u in [cfg.dim_mult[-1]] + cfg.dim_mult[::-1]]
    shortcut = []
    scale = 1.0
    inp = []
    blk = [("conv_in", "input_blocks.0.0", dict(cin=cfg.in_dim, cout=dim)),
           ("tt", "input_blocks.0.1", dict(c=dim, heads=cfg.num_heads, dh=cfg.head_dim))]
    inp.append(blk)
    shortcut.append(dim)
    idx = 1
    out_dim = dim
    for i, (in_dim, out_dim) in enumerate(zip(enc_dims[:-1], enc_dims[1:])):
        for j in range(cfg.num_res_blocks):
            p = f"input_blocks.{idx}"
            blk = [("res", f"{p}.0", dict(cin=in_dim, cout=out_dim))]
            if scale in cfg.attn_scales:
                blk.append(("st", f"{p}.1", dict(c=out_dim, heads=out_dim // cfg.head_dim, dh=cfg.head_dim,
                                                 ctx=cfg.context_dim)))
                blk.append(("tt", f"{p}.2", dict(c=out_dim, heads=out_dim // cfg.head_dim, dh=cfg.head_dim)))
            in_dim = out_dim
            inp.append(blk)
            shortcut.append(out_dim)
            idx += 1
            if i != len(cfg.dim_mult) - 1 and j == cfg.num_res_blocks - 1:
                inp.append([("down", f"input_blocks.{idx}", dict(c=out_dim))])
                shortcut.append(out_dim)
                scale /= 2.0
                idx += 1
    mid = [("res", "middle_block.0", dict(cin=out_dim, cout=out_dim)),
           ("st", "middle_block.1", dict(c=out_dim, heads=out_dim // cfg.head_dim, dh=cfg.head_dim,
                                         ctx=cfg.context_dim)),
           ("tt", "middle_block.2", dict(c=out_dim, heads=out_dim // cfg.head_dim, dh=cfg.head_dim)),
           ("res", "middle_block.3", dict(cin=out_dim, cout=out_dim))]
    outb = []
    idx = 0
    for i, (in_dim, out_dim) in enumerate(zip(dec_dims[:-1], dec_dims[1:])):
        for j in range(cfg.num_res_blocks + 1):
            p = f"output_blocks.{idx}"
            sc = shortcut.pop()
            blk = [("res", f"{p}.0", dict(cin=in_dim + sc, cout=out_dim, skip=sc))]
            k = 1
            if scale in cfg.attn_scales:
                blk.append(("st", f"{p}.{k}", dict(c=out_dim, heads=out_dim // cfg.head_dim, dh=cfg.head_dim,
                                                   ctx=cfg.dec_context_dim)))
                blk.append(("tt", f"{p}.{k + 1}", dict(c=out_dim, heads=out_dim // cfg.head_dim,
                                                       dh=cfg.head_dim)))
                k += 2
            in_dim = out_dim
            if i != len(cfg.dim_mult) - 1 and j == cfg.num_res_blocks:
                blk.append(("up", f"{p}.{k}", dict(c=out_dim)))
                scale *= 2.0
            outb.append(blk)
            idx += 1
    return inp, mid, outb


# ----------------------------------------------------------------------------- leaf restatements
def sinusoidal_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    half = dim // 2
    t = t.float()
    freq = torch.pow(10000, -torch.arange(half).to(t).div(half))
    s = torch.outer(t, freq)
    x = torch.cat([torch.cos(s), torch.sin(s)], dim=1)  # cos first (util.py:186)
    if dim % 2:
        x = torch.cat([x, torch.zeros_like(x[:, :1])], dim=1)
    return x


def _mlp(sd, p, x):
    x = F.linear(x, sd[f"{p}.0.weight"], sd[f"{p}.0.bias"])
    x = F.silu(x)
    return F.linear(x, sd[f"{p}.2.weight"], sd[f"{p}.2.bias"])


def attention(sd, p, x, context, heads):
    """x [b, n, c]; exact softmax attention, scale 1/sqrt(dh), no mask (util.py:230-268)."""
    ctx = x if context is None else context
    q = F.linear(x, sd[f"{p}.to_q.weight"])
    k = F.linear(ctx, sd[f"{p}.to_k.weight"])
    v = F.linear(ctx, sd[f"{p}.to_v.weight"])
    b, n, inner = q.shape
    dh = inner // heads

    def split(t):
        return t.reshape(b, t.shape[1], heads, dh).permute(0, 2, 1, 3)

    q, k, v = split(q), split(k), split(v)
    s = torch.matmul(q, k.transpose(-1, -2)) * (dh ** -0.5)
    o = torch.matmul(torch.softmax(s, dim=-1), v)
    o = o.permute(0, 2, 1, 3).reshape(b, n, inner)
    return F.linear(o, sd[f"{p}.to_out.0.weight"], sd[f"{p}.to_out.0.bias"])


def basic_transformer_block(sd, p, x, context, heads):
    def ln(name, t):
        return F.layer_norm(t, (t.shape[-1],), sd[f"{p}.{name}.weight"], sd[f"{p}.{name}.bias"], 1e-5)

    x = attention(sd, f"{p}.attn1", ln("norm1", x), None, heads) + x
    x = attention(sd, f"{p}.attn2", ln("norm2", x), context, heads) + x
    h = F.linear(ln("norm3", x), sd[f"{p}.ff.net.0.proj.weight"], sd[f"{p}.ff.net.0.proj.bias"])
    a, g = h.chunk(2, dim=-1)
    h = a * F.gelu(g)  # exact erf GELU
    h = F.linear(h, sd[f"{p}.ff.net.2.weight"], sd[f"{p}.ff.net.2.bias"])
    return h + x


def spatial_transformer(sd, p, x, context, heads):
    """x [(b f), c, h, w]; context [(b f), L, ctx]."""
    n, c, h, w = x.shape
    x_in = x
    x = F.group_norm(x, 32, sd[f"{p}.norm.weight"], sd[f"{p}.norm.bias"], 1e-6)
    x = x.permute(0, 2, 3, 1).reshape(n, h * w, c)
    x = F.linear(x, sd[f"{p}.proj_in.weight"], sd[f"{p}.proj_in.bias"])
    x = basic_transformer_block(sd, f"{p}.transformer_blocks.0", x, context, heads)
    x = F.linear(x, sd[f"{p}.proj_out.weight"], sd[f"{p}.proj_out.bias"])
    x = x.reshape(n, h, w, c).permute(0, 3, 1, 2)
    return x + x_in


def temporal_transformer(sd, p, x, heads):
    """x [b, c, f, h, w]; GN statistics span all frames; both attentions are self-attn."""
    b, c, f, h, w = x.shape
    x_in = x
    x = F.group_norm(x, 32, sd[f"{p}.norm.weight"], sd[f"{p}.norm.bias"], 1e-6)
    x = x.permute(0, 3, 4, 2, 1).reshape(b * h * w, f, c)  # (b h w) f c
    wi = sd[f"{p}.proj_in.weight"]
    x = F.linear(x, wi.reshape(wi.shape[0], wi.shape[1]), sd[f"{p}.proj_in.bias"])  # Conv1d k=1
    x = basic_transformer_block(sd, f"{p}.transformer_blocks.0", x, None, heads)
    wo = sd[f"{p}.proj_out.weight"]
    x = F.linear(x, wo.reshape(wo.shape[0], wo.shape[1]), sd[f"{p}.proj_out.bias"])
    x = x.reshape(b, h, w, f, c).permute(0, 4, 3, 1, 2)
    return x + x_in


def temporal_conv_block(sd, p, x):
    """x [b, c, f, h, w]: 4x [GN(5-D) -> SiLU -> Conv3d (3,1,1) zero-padded in f] + identity."""
    idn = x
    for name, ci in (("conv1", 2), ("conv2", 3), ("conv3", 3), ("conv4", 3)):
        x = F.group_norm(x, 32, sd[f"{p}.{name}.0.weight"], sd[f"{p}.{name}.0.bias"], 1e-5)
        x = F.silu(x)
        x = F.conv3d(x, sd[f"{p}.{name}.{ci}.weight"], sd[f"{p}.{name}.{ci}.bias"], padding=(1, 0, 0))
    return idn + x


def res_block(sd, p, x, emb, batch):
    """x [(b f), cin, h, w]; emb [(b f), E]."""
    h = F.group_norm(x, 32, sd[f"{p}.in_layers.0.weight"], sd[f"{p}.in_layers.0.bias"], 1e-5)
    h = F.conv2d(F.silu(h), sd[f"{p}.in_layers.2.weight"], sd[f"{p}.in_layers.2.bias"], padding=1)
    e = F.linear(F.silu(emb), sd[f"{p}.emb_layers.1.weight"], sd[f"{p}.emb_layers.1.bias"])
    h = h + e[:, :, None, None]
    h = F.group_norm(h, 32, sd[f"{p}.out_layers.0.weight"], sd[f"{p}.out_layers.0.bias"], 1e-5)
    h = F.conv2d(F.silu(h), sd[f"{p}.out_layers.3.weight"], sd[f"{p}.out_layers.3.bias"], padding=1)
    if f"{p}.skip_connection.weight" in sd:
        x = F.conv2d(x, sd[f"{p}.skip_connection.weight"], sd[f"{p}.skip_connection.bias"])
    h = x + h
    n, c, hh, ww = h.shape
    h5 = h.reshape(batch, n // batch, c, hh, ww).permute(0, 2, 1, 3, 4)
    h5 = temporal_conv_block(sd, f"{p}.temopral_conv", h5)
    return h5.permute(0, 2, 1, 3, 4).reshape(n, c, hh, ww)


def _run_block(sd, blk, x, emb, context, batch):
    for kind, p, m in blk:
        if kind == "conv_in":
            x = F.conv2d(x, sd[f"{p}.weight"], sd[f"{p}.bias"], padding=1)
        elif kind == "res":
            x = res_block(sd, p, x, emb, batch)
        elif kind == "st":
            x = spatial_transformer(sd, p, x, context, m["heads"])
        elif kind == "tt":
            n, c, h, w = x.shape
            x5 = x.reshape(batch, n // batch, c, h, w).permute(0, 2, 1, 3, 4)
            x5 = temporal_transformer(sd, p, x5, m["heads"])
            x = x5.permute(0, 2, 1, 3, 4).reshape(n, c, h, w)
        elif kind == "down":
            x = F.conv2d(x, sd[f"{p}.op.weight"], sd[f"{p}.op.bias"], stride=2, padding=1)
        elif kind == "up":
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            x = F.conv2d(x, sd[f"{p}.conv.weight"], sd[f"{p}.conv.bias"], padding=1)
        else:
            raise ValueError(kind)
    return x


@torch.no_grad()
def unet_forward(sd: Dict[str, torch.Tensor], cfg: UNetCfg, x: torch.Tensor, t: torch.Tensor,
                 y: torch.Tensor, camera_data: Optional[torch.Tensor] = None,
                 fps: Optional[torch.Tensor] = None, taps: Optional[dict] = None) -> torch.Tensor:
    """x [b, c, f, h, w] fp32, t [b] long, y [b, L, ctx], camera_data [b, f, 16] -> eps [b, out, f, h, w].

    ``taps`` (optional dict) receives the activation after every block (channels-first), keyed by the
    first prefix of the block — used by the per-block parity tests.
    """
    b, c, f, h, w = x.shape
    emb = _mlp(sd, "time_embed", sinusoidal_embedding(t, cfg.dim))
    if cfg.use_fps_condition and fps is not None:
        emb = emb + _mlp(sd, "fps_embedding", sinusoidal_embedding(fps, cfg.dim))
    emb = emb.repeat_interleave(f, dim=0)
    if cfg.use_camera_condition and camera_data is not None:
        emb = emb + _mlp(sd, "camera_embedding", camera_data.reshape(b * f, -1))
    context = y.repeat_interleave(f, dim=0)
    return unet_trunk(sd, cfg, x, emb, context, taps)


@torch.no_grad()
def unet_trunk(sd, cfg: UNetCfg, x, emb, context, taps=None):
    """Shared encoder / middle / decoder / head (unet_t2v.py:348-368 == unet_i2vgen.py:385-404).
    x [b, c_in, f, h, w]; emb [(b f), E]; context [(b f), L, ctx]."""
    b, c, f, h, w = x.shape
    x = x.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    inp, mid, outb = block_plan(cfg)
    xs = []
    for blk in inp:
        x = _run_block(sd, blk, x, emb, context, b)
        xs.append(x)
        if taps is not None:
            taps[blk[0][1]] = x
    x = _run_block(sd, mid, x, emb, context, b)
    if taps is not None:
        taps["middle_block"] = x
    for blk in outb:
        x = torch.cat([x, xs.pop()], dim=1)
        x = _run_block(sd, blk, x, emb, context, b)
        if taps is not None:
            taps[blk[0][1]] = x
    x = F.group_norm(x, 32, sd["out.0.weight"], sd["out.0.bias"], 1e-5)
    x = F.conv2d(F.silu(x), sd["out.2.weight"], sd["out.2.bias"], padding=1)
    return x.reshape(b, f, cfg.out_dim, h, w).permute(0, 2, 1, 3, 4)

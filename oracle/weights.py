"""ORACLE helper (test infrastructure): parameter manifests and deterministic synthetic weights.

No checkpoint ships with the reference and a freshly constructed reference model is almost
the identity because of its zero-initialised layers, so every parity test runs on seeded
random weights in which *all* tensors (including the zero-inits) are re-randomised.  The same generator
is used by ``make_golden.py`` (weights loaded into the imported reference) and by the tests (weights
given to the oracle / the HIP path), so fixtures only need to store inputs and outputs.
"""
import math
from collections import OrderedDict

import torch

from .unet_ref import UNetCfg, block_plan


def _attn_shapes(p, c, inner, ctx):
    kv = inner if ctx is None else ctx
    return [(f"{p}.to_q.weight", (inner, c)), (f"{p}.to_k.weight", (inner, kv)),
            (f"{p}.to_v.weight", (inner, kv)), (f"{p}.to_out.0.weight", (c, inner)),
            (f"{p}.to_out.0.bias", (c,))]


def _tblock_shapes(p, inner, ctx):
    s = []
    s += _attn_shapes(f"{p}.attn1", inner, inner, None)
    s += [(f"{p}.ff.net.0.proj.weight", (inner * 8, inner)), (f"{p}.ff.net.0.proj.bias", (inner * 8,)),
          (f"{p}.ff.net.2.weight", (inner, inner * 4)), (f"{p}.ff.net.2.bias", (inner,))]
    s += _attn_shapes(f"{p}.attn2", inner, inner, ctx)
    for n in ("norm1", "norm2", "norm3"):
        s += [(f"{p}.{n}.weight", (inner,)), (f"{p}.{n}.bias", (inner,))]
    return s


def unet_param_shapes(cfg: UNetCfg) -> "OrderedDict[str, tuple]":
    """State-dict key -> shape, in the reference's registration order (unet_t2v.py:141-265)."""
    E = cfg.dim * 4
    s = [("time_embed.0.weight", (E, cfg.dim)), ("time_embed.0.bias", (E,)),
         ("time_embed.2.weight", (E, E)), ("time_embed.2.bias", (E,))]
    if cfg.use_camera_condition:
        s += [("camera_embedding.0.weight", (E, cfg.camera_dim)), ("camera_embedding.0.bias", (E,)),
              ("camera_embedding.2.weight", (E, E)), ("camera_embedding.2.bias", (E,))]
    if cfg.use_fps_condition:
        s += [("fps_embedding.0.weight", (E, cfg.dim)), ("fps_embedding.0.bias", (E,)),
              ("fps_embedding.2.weight", (E, E)), ("fps_embedding.2.bias", (E,))]
    inp, mid, outb = block_plan(cfg)
    for blk in inp + [mid] + outb:
        for kind, p, m in blk:
            if kind == "conv_in":
                s += [(f"{p}.weight", (m["cout"], m["cin"], 3, 3)), (f"{p}.bias", (m["cout"],))]
            elif kind == "res":
                ci, co = m["cin"], m["cout"]
                s += [(f"{p}.in_layers.0.weight", (ci,)), (f"{p}.in_layers.0.bias", (ci,)),
                      (f"{p}.in_layers.2.weight", (co, ci, 3, 3)), (f"{p}.in_layers.2.bias", (co,)),
                      (f"{p}.emb_layers.1.weight", (co, E)), (f"{p}.emb_layers.1.bias", (co,)),
                      (f"{p}.out_layers.0.weight", (co,)), (f"{p}.out_layers.0.bias", (co,)),
                      (f"{p}.out_layers.3.weight", (co, co, 3, 3)), (f"{p}.out_layers.3.bias", (co,))]
                if ci != co:
                    s += [(f"{p}.skip_connection.weight", (co, ci, 1, 1)), (f"{p}.skip_connection.bias", (co,))]
                for name, idx in (("conv1", 2), ("conv2", 3), ("conv3", 3), ("conv4", 3)):
                    q = f"{p}.temopral_conv.{name}"
                    s += [(f"{q}.0.weight", (co,)), (f"{q}.0.bias", (co,)),
                          (f"{q}.{idx}.weight", (co, co, 3, 1, 1)), (f"{q}.{idx}.bias", (co,))]
            elif kind == "st":
                c = m["c"]
                inner = m["heads"] * m["dh"]
                s += [(f"{p}.norm.weight", (c,)), (f"{p}.norm.bias", (c,)),
                      (f"{p}.proj_in.weight", (inner, c)), (f"{p}.proj_in.bias", (inner,))]
                s += _tblock_shapes(f"{p}.transformer_blocks.0", inner, m["ctx"])
                s += [(f"{p}.proj_out.weight", (inner, c)), (f"{p}.proj_out.bias", (inner,))]
            elif kind == "tt":
                c = m["c"]
                inner = m["heads"] * m["dh"]
                s += [(f"{p}.norm.weight", (c,)), (f"{p}.norm.bias", (c,)),
                      (f"{p}.proj_in.weight", (inner, c, 1)), (f"{p}.proj_in.bias", (inner,))]
                s += _tblock_shapes(f"{p}.transformer_blocks.0", inner, None)
                s += [(f"{p}.proj_out.weight", (c, inner, 1)), (f"{p}.proj_out.bias", (c,))]
            elif kind == "down":
                s += [(f"{p}.op.weight", (m["c"], m["c"], 3, 3)), (f"{p}.op.bias", (m["c"],))]
            elif kind == "up":
                s += [(f"{p}.conv.weight", (m["c"], m["c"], 3, 3)), (f"{p}.conv.bias", (m["c"],))]
    s += [("out.0.weight", (cfg.dim,)), ("out.0.bias", (cfg.dim,)),
          ("out.2.weight", (cfg.out_dim, cfg.dim, 3, 3)), ("out.2.bias", (cfg.out_dim,))]
    return OrderedDict(s)


def vae_decoder_param_shapes(ch=128, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, z_channels=4,
                             embed_dim=4) -> "OrderedDict[str, tuple]":
    """Keys of ``AutoencoderKL.post_quant_conv`` + ``decoder.*`` (autoencoder.py:52, 582-652)."""
    s = [("post_quant_conv.weight", (z_channels, embed_dim, 1, 1)), ("post_quant_conv.bias", (z_channels,))]
    nres = len(ch_mult)
    block_in = ch * ch_mult[-1]
    s += [("decoder.conv_in.weight", (block_in, z_channels, 3, 3)), ("decoder.conv_in.bias", (block_in,))]

    def res(p, ci, co):
        r = [(f"{p}.norm1.weight", (ci,)), (f"{p}.norm1.bias", (ci,)),
             (f"{p}.conv1.weight", (co, ci, 3, 3)), (f"{p}.conv1.bias", (co,)),
             (f"{p}.norm2.weight", (co,)), (f"{p}.norm2.bias", (co,)),
             (f"{p}.conv2.weight", (co, co, 3, 3)), (f"{p}.conv2.bias", (co,))]
        if ci != co:
            r += [(f"{p}.nin_shortcut.weight", (co, ci, 1, 1)), (f"{p}.nin_shortcut.bias", (co,))]
        return r

    s += res("decoder.mid.block_1", block_in, block_in)
    p = "decoder.mid.attn_1"
    s += [(f"{p}.norm.weight", (block_in,)), (f"{p}.norm.bias", (block_in,))]
    for n in ("q", "k", "v", "proj_out"):
        s += [(f"{p}.{n}.weight", (block_in, block_in, 1, 1)), (f"{p}.{n}.bias", (block_in,))]
    s += res("decoder.mid.block_2", block_in, block_in)
    ups = []
    for i_level in reversed(range(nres)):
        block_out = ch * ch_mult[i_level]
        lvl = []
        for i_block in range(num_res_blocks + 1):
            lvl += res(f"decoder.up.{i_level}.block.{i_block}", block_in, block_out)
            block_in = block_out
        if i_level != 0:
            lvl += [(f"decoder.up.{i_level}.upsample.conv.weight", (block_in, block_in, 3, 3)),
                    (f"decoder.up.{i_level}.upsample.conv.bias", (block_in,))]
        ups.insert(0, lvl)  # the reference prepends (autoencoder.py:641) -> level 0 registers first
    for lvl in ups:
        s += lvl
    s += [("decoder.norm_out.weight", (block_in,)), ("decoder.norm_out.bias", (block_in,)),
          ("decoder.conv_out.weight", (out_ch, block_in, 3, 3)), ("decoder.conv_out.bias", (out_ch,))]
    return OrderedDict(s)


def vae_encoder_param_shapes(ch=128, in_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, z_channels=4, embed_dim=4):
    """Keys of ``encoder.*`` + ``quant_conv`` (autoencoder.py:51, 483-546)."""
    def res(p, ci, co):
        r = [(f"{p}.norm1.weight", (ci,)), (f"{p}.norm1.bias", (ci,)), (f"{p}.conv1.weight", (co, ci, 3, 3)),
             (f"{p}.conv1.bias", (co,)), (f"{p}.norm2.weight", (co,)), (f"{p}.norm2.bias", (co,)),
             (f"{p}.conv2.weight", (co, co, 3, 3)), (f"{p}.conv2.bias", (co,))]
        if ci != co:
            r += [(f"{p}.nin_shortcut.weight", (co, ci, 1, 1)), (f"{p}.nin_shortcut.bias", (co,))]
        return r
    s = [("encoder.conv_in.weight", (ch, in_ch, 3, 3)), ("encoder.conv_in.bias", (ch,))]
    in_mult = (1,) + tuple(ch_mult)
    block_in = ch
    for lvl in range(len(ch_mult)):
        block_in, block_out = ch * in_mult[lvl], ch * ch_mult[lvl]
        for i in range(num_res_blocks):
            s += res(f"encoder.down.{lvl}.block.{i}", block_in, block_out)
            block_in = block_out
        if lvl != len(ch_mult) - 1:
            s += [(f"encoder.down.{lvl}.downsample.conv.weight", (block_in, block_in, 3, 3)),
                  (f"encoder.down.{lvl}.downsample.conv.bias", (block_in,))]
    s += res("encoder.mid.block_1", block_in, block_in)
    p = "encoder.mid.attn_1"
    s += [(f"{p}.norm.weight", (block_in,)), (f"{p}.norm.bias", (block_in,))]
    for n in ("q", "k", "v", "proj_out"):
        s += [(f"{p}.{n}.weight", (block_in, block_in, 1, 1)), (f"{p}.{n}.bias", (block_in,))]
    s += res("encoder.mid.block_2", block_in, block_in)
    s += [("encoder.norm_out.weight", (block_in,)), ("encoder.norm_out.bias", (block_in,)),
          ("encoder.conv_out.weight", (2 * z_channels, block_in, 3, 3)), ("encoder.conv_out.bias", (2 * z_channels,)),
          ("quant_conv.weight", (2 * embed_dim, 2 * z_channels, 1, 1)), ("quant_conv.bias", (2 * embed_dim,))]
    return OrderedDict(s)


def random_state_dict(shapes, seed: int, gain: float = 1.0, dtype=torch.float32):
    """Deterministic synthetic weights: matrices/filters ~ N(0, gain/fan_in), norm scales ~ 1 + 0.1 N,
    biases ~ 0.05 N.  Every tensor is random (zero-inits re-randomised).  Values are rounded to
    bf16-representable numbers so that the bf16 HIP path and the fp32 oracle see *identical* weights."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for k, shp in shapes.items():
        if len(shp) == 1:
            if k.endswith(".bias"):
                v = 0.05 * torch.randn(shp, generator=g)
            else:
                v = 1.0 + 0.1 * torch.randn(shp, generator=g)
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            v = torch.randn(shp, generator=g) * math.sqrt(gain / fan_in)
        sd[k] = v.to(torch.bfloat16).to(dtype)
    return sd


def checksum(sd) -> float:
    """Order-dependent scalar fingerprint of a state dict (detects RNG / ordering drift)."""
    acc = 0.0
    for i, (k, v) in enumerate(sd.items()):
        acc += (i + 1) * float(v.double().abs().sum())
    return acc

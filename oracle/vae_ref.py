"""ORACLE (test infrastructure, CPU, fp32) — restatement of the reference SD-VAE decoder and encoder.

Follows ``AutoencoderKL.decode`` (tools/modules/autoencoder.py:101-104), ``Decoder.forward`` (:654-687),
``ResnetBlock.forward`` (:316-336), ``AttnBlock.forward`` (:366-390; single head, scale c^-0.5),
``Upsample.forward`` (:456-460; nearest x2 then conv) and ``Normalize`` (GroupNorm 32, eps 1e-6, :21-22).
Pinned by ``tests/golden/vae_tiny.safetensors``.
"""
import torch
import torch.nn.functional as F


def _gn(sd, p, x):
    return F.group_norm(x, 32, sd[f"{p}.weight"], sd[f"{p}.bias"], 1e-6)


def _swish(x):
    return x * torch.sigmoid(x)


def _res(sd, p, x):
    h = F.conv2d(_swish(_gn(sd, f"{p}.norm1", x)), sd[f"{p}.conv1.weight"], sd[f"{p}.conv1.bias"], padding=1)
    h = F.conv2d(_swish(_gn(sd, f"{p}.norm2", h)), sd[f"{p}.conv2.weight"], sd[f"{p}.conv2.bias"], padding=1)
    if f"{p}.nin_shortcut.weight" in sd:
        x = F.conv2d(x, sd[f"{p}.nin_shortcut.weight"], sd[f"{p}.nin_shortcut.bias"])
    return x + h


def _attn(sd, p, x):
    b, c, h, w = x.shape
    hn = _gn(sd, f"{p}.norm", x)
    q = F.conv2d(hn, sd[f"{p}.q.weight"], sd[f"{p}.q.bias"]).reshape(b, c, h * w).permute(0, 2, 1)
    k = F.conv2d(hn, sd[f"{p}.k.weight"], sd[f"{p}.k.bias"]).reshape(b, c, h * w)
    v = F.conv2d(hn, sd[f"{p}.v.weight"], sd[f"{p}.v.bias"]).reshape(b, c, h * w)
    w_ = torch.softmax(torch.bmm(q, k) * (int(c) ** -0.5), dim=2)
    o = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, h, w)
    return x + F.conv2d(o, sd[f"{p}.proj_out.weight"], sd[f"{p}.proj_out.bias"])


@torch.no_grad()
def vae_decode(sd, z, ch_mult=(1, 2, 4, 4), num_res_blocks=2):
    """z [n, 4, h, w] (already divided by the 0.18215 scale factor) -> image [n, 3, 8h, 8w]."""
    z = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    h = F.conv2d(z, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    h = _res(sd, "decoder.mid.block_1", h)
    h = _attn(sd, "decoder.mid.attn_1", h)
    h = _res(sd, "decoder.mid.block_2", h)
    for lvl in reversed(range(len(ch_mult))):
        for i in range(num_res_blocks + 1):
            h = _res(sd, f"decoder.up.{lvl}.block.{i}", h)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, sd[f"decoder.up.{lvl}.upsample.conv.weight"],
                         sd[f"decoder.up.{lvl}.upsample.conv.bias"], padding=1)
    h = _swish(_gn(sd, "decoder.norm_out", h))
    return F.conv2d(h, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)


@torch.no_grad()
def vae_encode_moments(sd, x, ch_mult=(1, 2, 4, 4), num_res_blocks=2):
    """image [n, 3, H, W] -> moments [n, 2*zc, H/8, W/8] = quant_conv(Encoder(x)) (autoencoder.py:76-80, Encoder.forward
    :548-576; Downsample = pad (0,1,0,1) then valid stride-2 conv, :475-479)."""
    h = F.conv2d(x, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    for lvl in range(len(ch_mult)):
        for i in range(num_res_blocks):
            h = _res(sd, f"encoder.down.{lvl}.block.{i}", h)
        if lvl != len(ch_mult) - 1:
            h = F.pad(h, (0, 1, 0, 1))
            h = F.conv2d(h, sd[f"encoder.down.{lvl}.downsample.conv.weight"], sd[f"encoder.down.{lvl}.downsample.conv.bias"],
                         stride=2)
    h = _res(sd, "encoder.mid.block_1", h)
    h = _attn(sd, "encoder.mid.attn_1", h)
    h = _res(sd, "encoder.mid.block_2", h)
    h = F.conv2d(_swish(_gn(sd, "encoder.norm_out", h)), sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)
    return F.conv2d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])


def posterior_sample(moments, noise, scale_factor=1.0):
    """DiagonalGaussianDistribution.sample() * scale_factor (autoencoder.py:213-226, :19-28) with the noise given."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    return scale_factor * (mean + torch.exp(0.5 * logvar.clamp(-30.0, 20.0)) * noise)

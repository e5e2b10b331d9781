"""ORACLE (test infrastructure, CPU, fp32) — restatement of the reference's LGM branch up to the Gaussians.

Follows ``core/unet.py`` (``ResnetBlock.forward`` :84-101, ``MVAttention.forward`` :36-51, ``DownBlock`` :136-150,
``MidBlock`` :176-182, ``UpBlock`` :214-230, ``UNet.forward`` :289-319), ``core/attention.py:44-57`` (scaled exact softmax
attention over the 4 x H x W tokens of a sample, qkv without bias), ``core/models.py:87-113`` (``forward_gaussians``:
U-Net -> 1x1 conv -> [B, V*S*S, 14] -> pos clamp / sigmoid opacity / 0.1*softplus scale / F.normalize(dim=1) rotation /
0.5*tanh+0.5 colour), ``core/utils.py:10-43`` (``get_rays``) and the camera preparation of
``tools/inferences/inference_text2video_entrance.py:198-235``.
Pinned by ``tests/golden/lgm_unet_tiny.safetensors`` / ``lgm_rays.safetensors`` (generated from the imported reference).
"""
import dataclasses
import math
from typing import Tuple

import torch
import torch.nn.functional as F


@dataclasses.dataclass
class LgmCfg:
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
    # options of config_defaults['big'] that the inference path reads (core/options.py:8-19,73-83)
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


def lgm_unet_param_shapes(cfg: LgmCfg):
    """state-dict manifest of core.unet.UNet in registration order (checked against the reference in make_golden)."""
    s = [("conv_in.weight", (cfg.down_channels[0], cfg.in_channels, 3, 3)), ("conv_in.bias", (cfg.down_channels[0],))]
    cout = cfg.down_channels[0]
    nd = len(cfg.down_channels)
    for i in range(nd):
        cin, cout = cout, cfg.down_channels[i]
        for j in range(cfg.layers_per_block):
            s += _res_shapes(f"down_blocks.{i}.nets.{j}", cin if j == 0 else cout, cout)
        if cfg.down_attention[i]:
            for j in range(cfg.layers_per_block):
                s += _attn_shapes(f"down_blocks.{i}.attns.{j}", cout)
        if i != nd - 1:
            s += [(f"down_blocks.{i}.downsample.weight", (cout, cout, 3, 3)), (f"down_blocks.{i}.downsample.bias", (cout,))]
    cm = cfg.down_channels[-1]
    s += _res_shapes("mid_block.nets.0", cm, cm) + _res_shapes("mid_block.nets.1", cm, cm)
    if cfg.mid_attention:
        s += _attn_shapes("mid_block.attns.0", cm)
    cout = cfg.up_channels[0]
    nu = len(cfg.up_channels)
    for i in range(nu):
        cin, cout = cout, cfg.up_channels[i]
        cskip = cfg.down_channels[max(-2 - i, -nd)]
        nl = cfg.layers_per_block + 1
        for j in range(nl):
            ci = cin if j == 0 else cout
            cs = cskip if j == nl - 1 else cout
            s += _res_shapes(f"up_blocks.{i}.nets.{j}", ci + cs, cout)
        if cfg.up_attention[i]:
            for j in range(nl):
                s += _attn_shapes(f"up_blocks.{i}.attns.{j}", cout)
        if i != nu - 1:
            s += [(f"up_blocks.{i}.upsample.weight", (cout, cout, 3, 3)), (f"up_blocks.{i}.upsample.bias", (cout,))]
    s += [("norm_out.weight", (cfg.up_channels[-1],)), ("norm_out.bias", (cfg.up_channels[-1],)),
          ("conv_out.weight", (cfg.out_channels, cfg.up_channels[-1], 3, 3)), ("conv_out.bias", (cfg.out_channels,))]
    return dict(s)


def _res(sd, p, x, k):
    h = F.conv2d(F.silu(F.group_norm(x, 32, sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"], 1e-5)),
                 sd[f"{p}.conv1.weight"], sd[f"{p}.conv1.bias"], padding=1)
    h = F.conv2d(F.silu(F.group_norm(h, 32, sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"], 1e-5)),
                 sd[f"{p}.conv2.weight"], sd[f"{p}.conv2.bias"], padding=1)
    if f"{p}.shortcut.weight" in sd:
        x = F.conv2d(x, sd[f"{p}.shortcut.weight"], sd[f"{p}.shortcut.bias"])
    return (h + x) * k


def _attn(sd, p, x, heads, frames, k):
    BV, C, H, W = x.shape
    B = BV // frames
    h = F.group_norm(x, 32, sd[f"{p}.norm.weight"], sd[f"{p}.norm.bias"], 1e-5)
    tok = h.reshape(B, frames, C, H, W).permute(0, 1, 3, 4, 2).reshape(B, -1, C)
    N = tok.shape[1]
    qkv = (tok @ sd[f"{p}.attn.qkv.weight"].t()).reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    o = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])          # scale = head_dim ** -0.5
    o = o.transpose(1, 2).reshape(B, N, C) @ sd[f"{p}.attn.proj.weight"].t() + sd[f"{p}.attn.proj.bias"]
    o = o.reshape(B, frames, H, W, C).permute(0, 1, 4, 2, 3).reshape(BV, C, H, W)
    return (o + x) * k


@torch.no_grad()
def lgm_unet_forward(sd, cfg: LgmCfg, x, taps=None):
    """x [B*V, in, H, W] -> [B*V, out, H', W']; taps (optional dict) collects the block outputs."""
    k, nd, nu = cfg.skip_scale, len(cfg.down_channels), len(cfg.up_channels)
    x = F.conv2d(x, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    xss = [x]
    for i in range(nd):
        for j in range(cfg.layers_per_block):
            x = _res(sd, f"down_blocks.{i}.nets.{j}", x, k)
            if cfg.down_attention[i]:
                x = _attn(sd, f"down_blocks.{i}.attns.{j}", x, cfg.num_heads, cfg.num_frames, k)
            xss.append(x)
        if i != nd - 1:
            x = F.conv2d(x, sd[f"down_blocks.{i}.downsample.weight"], sd[f"down_blocks.{i}.downsample.bias"], stride=2, padding=1)
            xss.append(x)
        if taps is not None:
            taps[f"down_blocks.{i}"] = x
    x = _res(sd, "mid_block.nets.0", x, k)
    if cfg.mid_attention:
        x = _attn(sd, "mid_block.attns.0", x, cfg.num_heads, cfg.num_frames, k)
    x = _res(sd, "mid_block.nets.1", x, k)
    if taps is not None:
        taps["mid_block"] = x
    for i in range(nu):
        nl = cfg.layers_per_block + 1
        xs, xss = xss[-nl:], xss[:-nl]
        for j in range(nl):
            x = _res(sd, f"up_blocks.{i}.nets.{j}", torch.cat([x, xs[-1]], dim=1), k)
            xs = xs[:-1]
            if cfg.up_attention[i]:
                x = _attn(sd, f"up_blocks.{i}.attns.{j}", x, cfg.num_heads, cfg.num_frames, k)
        if i != nu - 1:
            x = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), sd[f"up_blocks.{i}.upsample.weight"],
                         sd[f"up_blocks.{i}.upsample.bias"], padding=1)
        if taps is not None:
            taps[f"up_blocks.{i}"] = x
    x = F.silu(F.group_norm(x, 32, sd["norm_out.weight"], sd["norm_out.bias"], 1e-5))
    return F.conv2d(x, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)


def gaussian_activation(x):
    """[B, N, 14] raw -> (pos clamp[-1,1], sigmoid opacity, 0.1*softplus scale, rotation, 0.5*tanh+0.5 rgb).
    The reference's ``rot_act = F.normalize`` is applied with its DEFAULT dim=1 to a [B, N, 4] tensor (core/models.py:43,
    :108): every quaternion COMPONENT is divided by its L2 norm over the N Gaussians — not a per-Gaussian unit
    quaternion.  Kept as is (behaviour = API)."""
    pos = x[..., 0:3].clamp(-1, 1)
    opacity = torch.sigmoid(x[..., 3:4])
    scale = 0.1 * F.softplus(x[..., 4:7])
    rot = F.normalize(x[..., 7:11], dim=1)
    rgb = 0.5 * torch.tanh(x[..., 11:]) + 0.5
    return torch.cat([pos, opacity, scale, rot, rgb], dim=-1)


@torch.no_grad()
def forward_gaussians(sd, cfg: LgmCfg, images):
    """images [B, V, 9, H, W] -> gaussians [B, V*S*S, 14]; sd holds ``unet.*`` and ``conv.*`` (LGM's key names)."""
    B, V, C, H, W = images.shape
    usd = {k[len("unet."):]: v for k, v in sd.items() if k.startswith("unet.")}
    x = lgm_unet_forward(usd, cfg, images.reshape(B * V, C, H, W))
    x = F.conv2d(x, sd["conv.weight"], sd["conv.bias"])
    S = x.shape[-1]
    x = x.reshape(B, V, 14, S, S).permute(0, 1, 3, 4, 2).reshape(B, -1, 14)
    return gaussian_activation(x)


def get_rays(pose, h, w, fovy):
    """OpenGL-convention rays of a camera-to-world pose: origins [h,w,3], unit directions [h,w,3]."""
    x, y = torch.meshgrid(torch.arange(w), torch.arange(h), indexing="xy")
    x, y = x.flatten().to(pose.dtype), y.flatten().to(pose.dtype)
    focal = h * 0.5 / math.tan(0.5 * math.radians(fovy))
    dirs = torch.stack([(x - w * 0.5 + 0.5) / focal, -(y - h * 0.5 + 0.5) / focal, -torch.ones_like(x)], dim=-1)
    rays_d = dirs @ pose[:3, :3].transpose(0, 1)
    rays_o = pose[:3, 3].unsqueeze(0).expand_as(rays_d)
    rays_d = rays_d / torch.sqrt(torch.clamp((rays_d * rays_d).sum(-1, keepdim=True), min=1e-20))
    return rays_o.reshape(h, w, 3), rays_d.reshape(h, w, 3)


def gs_data_from_camera(camera_data, cfg: LgmCfg = LgmCfg()):
    """camera_data [1, T, 16] (the UNet's camera condition) -> dict(input [1,T,6,256,256] Pluecker rays, cam_view,
    cam_view_proj [1,T,4,4], cam_pos [1,T,3]) — inference_text2video_entrance.py:198-235."""
    T = camera_data.shape[1]
    cam = camera_data.clone().reshape(T, 4, 4).contiguous().float()
    cam[:, 1] *= -1
    cam[:, [1, 2]] = cam[:, [2, 1]]
    cam[:, :3, 1:3] *= -1
    dist = float(torch.sqrt(cam[0, 0, 3] ** 2 + cam[0, 1, 3] ** 2 + cam[0, 2, 3] ** 2))
    transform = torch.tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, dist], [0, 0, 0, 1]], dtype=torch.float32) @ torch.inverse(cam[0])
    poses = transform.unsqueeze(0) @ cam
    rays = []
    for i in range(T):
        o, d = get_rays(poses[i], cfg.input_size, cfg.input_size, cfg.fovy)
        rays.append(torch.cat([torch.cross(o, d, dim=-1), d], dim=-1))
    rays = torch.stack(rays, dim=0).permute(0, 3, 1, 2).contiguous()
    tan = math.tan(0.5 * math.radians(cfg.fovy))
    proj = torch.zeros(4, 4)
    proj[0, 0] = proj[1, 1] = 1 / tan
    proj[2, 2] = (cfg.zfar + cfg.znear) / (cfg.zfar - cfg.znear)
    proj[3, 2] = -(cfg.zfar * cfg.znear) / (cfg.zfar - cfg.znear)
    proj[2, 3] = 1
    poses = poses.clone()
    poses[:, :3, 1:3] *= -1
    view = torch.inverse(poses).transpose(1, 2)
    return dict(input=rays.unsqueeze(0), cam_view=view.unsqueeze(0), cam_view_proj=(view @ proj).unsqueeze(0),
                cam_pos=(-poses[:, :3, 3]).unsqueeze(0))


@torch.no_grad()
def lgm_latent_z(lgm_sd, cfg: LgmCfg, vae_sd, z4, rays4, cam_view, cam_view_proj, noise, bg=0.5, scale_factor=0.18215,
                 vae_kw=None):
    """The whole ``autoencoder is not None`` branch of the video UNet for one CFG branch (unet_t2v.py:404-433), composed from
    the oracle pieces: z4 [4,C,h,w] (predicted x0 / 0.18215 of the 4 input views) -> VAE decode -> [0,1] clamp, ImageNet
    normalise, cat with the Pluecker rays -> Gaussians -> render all views (bg colour ``bg``), clamp -> nearest /2 ->
    (x-0.5)/0.5 -> VAE encode moments -> ``scale_factor * (mean + std * noise)`` -> latent_z [1,C,T,h,w]."""
    from .vae_ref import vae_decode, vae_encode_moments, posterior_sample
    from .gs_ref import render_views
    vae_kw = vae_kw or {}
    dec = vae_decode(vae_sd, z4, **vae_kw)
    x = (dec * 0.5 + 0.5).clamp(0, 1)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    images = torch.cat([(x - mean) / std, rays4], dim=1).unsqueeze(0)
    gauss = forward_gaussians(lgm_sd, cfg, images)[0]
    imgs, _ = render_views(gauss, cam_view, cam_view_proj, cfg.output_size, cfg.fovy, torch.full((3,), float(bg)))
    small = (imgs[:, :, ::2, ::2] - 0.5) / 0.5
    z = posterior_sample(vae_encode_moments(vae_sd, small, **vae_kw), noise, scale_factor)
    return z.unsqueeze(0).permute(0, 2, 1, 3, 4).contiguous()

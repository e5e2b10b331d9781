"""ORACLE (test infrastructure, CPU, fp32) — restatement of the OpenCLIP image tower (``model.encode_image``,
tools/modules/clip_embedder.py:187) = ``open_clip``'s ``VisionTransformer.forward`` for ViT-H/14: patch convolution (no bias),
class token, positional embedding, ``ln_pre``, the residual attention blocks (no mask), ``ln_post`` on the class token, ``@ proj``.

Only ``tests/`` (and the smoke / cpu_baseline legs) may import this module.  PARITY UNPINNED against the package (absent from
/root/reference and this image; no golden vectors in the reference's tree); ``tests/test_clip_cpu.py`` pins the block against
``torch.nn.MultiheadAttention`` / ``nn.LayerNorm`` / ``nn.GELU`` and the patch embedding against ``torch.nn.functional.conv2d``.
"""
import torch
import torch.nn.functional as F


def image_tower(sd, image, width, heads, layers, patch, taps=None):
    """sd: open_clip state dict (``visual.*`` keys, fp32); image [B, 3, S, S] -> [B, embed_dim]."""
    v = lambda k: sd["visual." + k].float()
    B = image.shape[0]
    d = width // heads
    x = F.conv2d(image.float(), v("conv1.weight"), stride=patch)                       # [B, W, g, g]
    x = x.reshape(B, width, -1).permute(0, 2, 1)
    x = torch.cat([v("class_embedding").expand(B, 1, width), x], dim=1) + v("positional_embedding")
    x = F.layer_norm(x, (width,), v("ln_pre.weight"), v("ln_pre.bias"), 1e-5)
    T = x.shape[1]
    for i in range(layers):
        p = f"transformer.resblocks.{i}."
        h = F.layer_norm(x, (width,), v(p + "ln_1.weight"), v(p + "ln_1.bias"), 1e-5)
        qkv = h @ v(p + "attn.in_proj_weight").t() + v(p + "attn.in_proj_bias")
        q, k, vv = (t.view(B, T, heads, d).transpose(1, 2) for t in qkv.split(width, dim=-1))
        a = (torch.softmax((q * d ** -0.5) @ k.transpose(-1, -2), dim=-1) @ vv).transpose(1, 2).reshape(B, T, width)
        x = x + a @ v(p + "attn.out_proj.weight").t() + v(p + "attn.out_proj.bias")
        h = F.layer_norm(x, (width,), v(p + "ln_2.weight"), v(p + "ln_2.bias"), 1e-5)
        h = F.gelu(h @ v(p + "mlp.c_fc.weight").t() + v(p + "mlp.c_fc.bias"))
        x = x + h @ v(p + "mlp.c_proj.weight").t() + v(p + "mlp.c_proj.bias")
        if taps is not None:
            taps[f"resblocks.{i}"] = x
    pooled = F.layer_norm(x[:, 0], (width,), v("ln_post.weight"), v("ln_post.bias"), 1e-5)
    return pooled @ v("proj")

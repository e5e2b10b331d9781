"""ORACLE (test infrastructure, CPU, fp32) — restatement of ``UNetSD_I2VGen.forward`` (tools/modules/unet/unet_i2vgen.py:
287-404): the I2VGen-XL image-conditioned front-end on top of the shared trunk of ``oracle/unet_ref.py``.

  * local_image_concat / local_temporal_encoder (``TransformerV2``: PreNorm attention + FeedForward, util.py:1091-1148,
    560-576) -> ``concat`` [b, 4, f, h, w], ADDED TWICE (:345-346, an acknowledged reference bug that is API);
  * local_image_embedding (:156-162) -> 64 context tokens; context_embedding(image) (:120-123) -> num_tokens tokens;
  * embeddings = time_embed + fps_embedding (+ camera) (:349-358); x = cat(x, concat) (:383).
Pinned by tests/golden/unet_i2v_tiny.safetensors (captured from the imported reference)."""
import dataclasses

import torch
import torch.nn.functional as F

from .unet_ref import UNetCfg, unet_trunk, sinusoidal_embedding, _mlp


def i2v_param_shapes(cfg: UNetCfg, concat_dim=4, num_tokens=4, y_dim=1024):
    """Extra keys of UNetSD_I2VGen on top of the T2V trunk (whose input conv takes in_dim + concat_dim channels)."""
    E, cd = cfg.dim * 4, concat_dim
    s = [("context_embedding.0.weight", (E, y_dim)), ("context_embedding.0.bias", (E,)),
         ("context_embedding.2.weight", (cfg.context_dim * num_tokens, E)), ("context_embedding.2.bias", (cfg.context_dim * num_tokens,)),
         ("fps_embedding.0.weight", (E, cfg.dim)), ("fps_embedding.0.bias", (E,)),
         ("fps_embedding.2.weight", (E, E)), ("fps_embedding.2.bias", (E,)),
         ("local_image_concat.0.weight", (cd * 4, 4, 3, 3)), ("local_image_concat.0.bias", (cd * 4,)),
         ("local_image_concat.2.weight", (cd * 4, cd * 4, 3, 3)), ("local_image_concat.2.bias", (cd * 4,)),
         ("local_image_concat.4.weight", (cd, cd * 4, 3, 3)), ("local_image_concat.4.bias", (cd,)),
         ("local_temporal_encoder.layers.0.0.norm.weight", (cd,)), ("local_temporal_encoder.layers.0.0.norm.bias", (cd,)),
         ("local_temporal_encoder.layers.0.0.fn.to_qkv.weight", (2 * cd * 3, cd)),
         ("local_temporal_encoder.layers.0.0.fn.to_out.0.weight", (cd, 2 * cd)),
         ("local_temporal_encoder.layers.0.0.fn.to_out.0.bias", (cd,)),
         ("local_temporal_encoder.layers.0.1.net.0.0.weight", (cd * 4, cd)), ("local_temporal_encoder.layers.0.1.net.0.0.bias", (cd * 4,)),
         ("local_temporal_encoder.layers.0.1.net.2.weight", (cd, cd * 4)), ("local_temporal_encoder.layers.0.1.net.2.bias", (cd,)),
         ("local_image_embedding.0.weight", (cd * 8, 4, 3, 3)), ("local_image_embedding.0.bias", (cd * 8,)),
         ("local_image_embedding.3.weight", (cd * 16, cd * 8, 3, 3)), ("local_image_embedding.3.bias", (cd * 16,)),
         ("local_image_embedding.5.weight", (1024, cd * 16, 3, 3)), ("local_image_embedding.5.bias", (1024,))]
    return dict(s)


def temporal_adapter(sd, x, heads=2):
    """TransformerV2 depth 1 on x [(b h w), f, cd]: x = attn(LN(x)) + x ; x = FF(x) + x  (util.py:1139-1143)."""
    p = "local_temporal_encoder.layers.0"
    n, f, cd = x.shape
    hn = F.layer_norm(x, (cd,), sd[f"{p}.0.norm.weight"], sd[f"{p}.0.norm.bias"], 1e-5)
    qkv = F.linear(hn, sd[f"{p}.0.fn.to_qkv.weight"])
    q, k, v = qkv.chunk(3, dim=-1)
    dh = q.shape[-1] // heads

    def split(t):
        return t.reshape(n, f, heads, dh).permute(0, 2, 1, 3)
    q, k, v = split(q), split(k), split(v)
    a = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * dh ** -0.5, dim=-1)
    o = torch.matmul(a, v).permute(0, 2, 1, 3).reshape(n, f, heads * dh)
    x = F.linear(o, sd[f"{p}.0.fn.to_out.0.weight"], sd[f"{p}.0.fn.to_out.0.bias"]) + x
    h = F.gelu(F.linear(x, sd[f"{p}.1.net.0.0.weight"], sd[f"{p}.1.net.0.0.bias"]))
    return F.linear(h, sd[f"{p}.1.net.2.weight"], sd[f"{p}.1.net.2.bias"]) + x


def i2v_concat(sd, local_image, f):
    """local_image [b, 4, h, w] (first-frame latent) -> concat [b, cd, f, h, w] (already doubled, :345-346)."""
    b, c, h, w = local_image.shape
    frames = [local_image]
    for tpos in range(f - 1):
        frames.append(torch.full_like(local_image, (tpos + 1) / (f - 1)))
    x = torch.stack(frames, dim=1).reshape(b * f, c, h, w) if f > 1 else local_image
    x = F.conv2d(x, sd["local_image_concat.0.weight"], sd["local_image_concat.0.bias"], padding=1)
    x = F.conv2d(F.silu(x), sd["local_image_concat.2.weight"], sd["local_image_concat.2.bias"], padding=1)
    x = F.conv2d(F.silu(x), sd["local_image_concat.4.weight"], sd["local_image_concat.4.bias"], padding=1)
    cd = x.shape[1]
    x = x.reshape(b, f, cd, h, w).permute(0, 3, 4, 1, 2).reshape(b * h * w, f, cd)      # (b h w) f c
    x = temporal_adapter(sd, x)
    x = x.reshape(b, h, w, f, cd).permute(0, 4, 3, 1, 2)
    return 2.0 * x


def i2v_local_tokens(sd, local_image):
    """local_image [b, 4, h, w] -> [b, 64, 1024] (:370-374)."""
    x = F.silu(F.conv2d(local_image, sd["local_image_embedding.0.weight"], sd["local_image_embedding.0.bias"], padding=1))
    x = F.adaptive_avg_pool2d(x, (32, 32))
    x = F.silu(F.conv2d(x, sd["local_image_embedding.3.weight"], sd["local_image_embedding.3.bias"], stride=2, padding=1))
    x = F.conv2d(x, sd["local_image_embedding.5.weight"], sd["local_image_embedding.5.bias"], stride=2, padding=1)
    b, c, hh, ww = x.shape
    return x.permute(0, 2, 3, 1).reshape(b, hh * ww, c)


@torch.no_grad()
def unet_i2v_forward(sd, cfg: UNetCfg, x, t, y, image, local_image, fps, camera_data=None, num_tokens=4, taps=None):
    """x [b,4,f,h,w]; y [b,L,ctx]; image [b,1,y_dim] or None; local_image [b,4,h,w]; fps [b] -> v/eps [b,out,f,h,w]."""
    b, c, f, h, w = x.shape
    concat = i2v_concat(sd, local_image, f)
    emb = _mlp(sd, "time_embed", sinusoidal_embedding(t, cfg.dim)) + _mlp(sd, "fps_embedding", sinusoidal_embedding(fps, cfg.dim))
    emb = emb.repeat_interleave(f, dim=0)
    if cfg.use_camera_condition and camera_data is not None:
        emb = emb + _mlp(sd, "camera_embedding", camera_data.reshape(b * f, -1))
    ctx = [y, i2v_local_tokens(sd, local_image)]
    if image is not None:
        ctx.append(_mlp(sd, "context_embedding", image).reshape(-1, num_tokens, cfg.context_dim))
    context = torch.cat(ctx, dim=1)
    if taps is not None:
        taps["concat"], taps["context"] = concat, context
    tcfg = dataclasses.replace(cfg, in_dim=cfg.in_dim + concat.shape[1])
    return unet_trunk(sd, tcfg, torch.cat([x, concat], dim=1), emb, context.repeat_interleave(f, dim=0), taps)

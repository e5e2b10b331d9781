"""Sampling pipeline of the t2v entrance (tools/inferences/inference_text2video_entrance.py:249-289) on the HIP
path: DDIM loop with batched CFG, then VAE decode of the 24 views in chunks of ``decoder_bs`` frames."""
import torch


@torch.no_grad()
def sample_views(unet, diffusion, autoencoder, noise, y_cond, y_uncond, camera_data, guide_scale=9.0,
                 ddim_timesteps=50, decoder_bs=4, scale_factor=0.18215, decode=True):
    """noise [1, 4, F, h, w] (device) -> (latent x0 [1,4,F,h,w], video [1, 3, F, 8h, 8w] in [-1, 1] or None).

    Mirrors the reference call (``diffusion.ddim_sample_loop(noise=..., model=..., model_kwargs=[cond, uncond],
    guide_scale=9.0, ddim_timesteps=50, eta=0.0)`` at :259-265) followed by ``1/scale_factor * z`` and the chunked
    ``autoencoder.decode`` of :280-289."""
    kw = [dict(y=y_cond, camera_data=camera_data), dict(y=y_uncond, camera_data=camera_data)]
    x0 = diffusion.ddim_sample_loop(noise=noise, model=unet, model_kwargs=kw, guide_scale=guide_scale,
                                    ddim_timesteps=ddim_timesteps, eta=0.0)
    if not decode:
        return x0, None
    comm = getattr(getattr(unet, "module", unet), "frame_comm", None)
    if comm is None:
        return x0, decode_views(autoencoder, x0, decoder_bs, scale_factor)
    # frame-parallel: the VAE is per-frame, so every rank decodes its own views; one gather of the images
    from .unet_t2v import gather_frames
    fl = x0.shape[2] // comm.world
    mine = decode_views(autoencoder, x0[:, :, comm.rank * fl:(comm.rank + 1) * fl].contiguous(), decoder_bs, scale_factor)
    return x0, gather_frames(comm, mine)


@torch.no_grad()
def decode_views(autoencoder, x0, decoder_bs=4, scale_factor=0.18215):
    """latent [b, 4, F, h, w] -> video [b, 3, F, 8h, 8w]: ``1/scale_factor * z`` then ``autoencoder.decode`` on chunks of
    ``decoder_bs`` frames (inference_text2video_entrance.py:280-289)."""
    b, c, f, h, w = x0.shape
    z = (x0 * (1.0 / scale_factor)).permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)   # b c f h w -> (b f) c h w
    # The reference decodes 4 frames at a time to bound its activation memory.  The HIP decoder is per-frame independent
    # (per-frame GroupNorm / attention) and the MI355X has 288 GB: it takes every frame in ONE plan when the top level
    # stays within the kernels' 32-bit row offsets (<= 4 M output pixels) — 6x fewer launches, fuller tile grids at the
    # low-resolution levels; the images are the same up to fp32 summation order inside a GEMM tile.
    # (b > 1 samples — `prompt_batch` — : the largest divisor of b * f that fits, e.g. one 24-frame plan per sample at 320 x 512)
    if getattr(autoencoder, "frame_independent", False):
        cap = (1 << 22) // (64 * h * w)
        fit = [d for d in range(1, b * f + 1) if (b * f) % d == 0 and d <= cap]
        if fit:
            decoder_bs = max(decoder_bs, fit[-1])
    outs = []
    for i in range(0, b * f, decoder_bs):
        outs.append(autoencoder.decode(z[i:i + decoder_bs].contiguous()))
    img = torch.cat(outs, dim=0)
    return img.reshape(b, f, img.shape[1], img.shape[2], img.shape[3]).permute(0, 2, 1, 3, 4).contiguous()

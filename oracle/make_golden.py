"""Generate ``tests/golden/*`` from the IMPORTED reference (authoring container only).

    python -m oracle.make_golden

Runs the reference modules (imported from /root/reference through ``oracle/shim.py``) on seeded inputs
with the synthetic weights of ``oracle/weights.py`` and stores inputs + expected outputs as small
safetensors fixtures.  The fixtures are data (tensors and key/shape manifests); no reference source is
copied.  ``tests/test_oracle_golden.py`` then pins the oracle restatement against them.
"""
import json
import os

import torch
from safetensors.torch import save_file

from . import shim
from .weights import (random_state_dict, unet_param_shapes, vae_decoder_param_shapes, checksum)
from .unet_ref import UNetCfg

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

TINY_A = dict(in_dim=4, dim=32, context_dim=1024, out_dim=4, dim_mult=[1, 2], num_heads=8, head_dim=32,
              num_res_blocks=1, attn_scales=[1.0, 0.5])
TINY_B = dict(in_dim=4, dim=64, context_dim=1024, out_dim=4, dim_mult=[1, 2, 2], num_heads=4, head_dim=64,
              num_res_blocks=2, attn_scales=[1.0, 0.5])


def build_ref_unet(ns, c: dict):
    U = ns.unet_t2v.UNetSD_T2VBase
    m = U(in_dim=c["in_dim"], dim=c["dim"], y_dim=c["context_dim"], context_dim=c["context_dim"],
          out_dim=c["out_dim"], dim_mult=c["dim_mult"], num_heads=c["num_heads"], head_dim=c["head_dim"],
          num_res_blocks=c["num_res_blocks"], attn_scales=c["attn_scales"], dropout=0.1,
          temporal_attention=True, use_checkpoint=False, use_camera_condition=True,
          use_fps_condition=False, use_lgm_refine=False)
    return m.eval()


def unet_case(ns, name, c, seed, F_, H, W, L):
    cfg = UNetCfg(**c)
    shapes = unet_param_shapes(cfg)
    ref = build_ref_unet(ns, c)
    ref_sd = ref.state_dict()
    assert list(ref_sd.keys()) == list(shapes.keys()), "manifest order mismatch"
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), (k, v.shape, shapes[k])
    sd = random_state_dict(shapes, seed)
    ref.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(seed + 1)
    B = 2
    x = torch.randn(B, 4, F_, H, W, generator=g)
    t = torch.tensor([501, 21])
    y = torch.randn(B, L, c["context_dim"], generator=g)
    cam = torch.randn(B, F_, 16, generator=g)
    taps = {}

    def hook(key):
        def fn(mod, inp, out):
            taps[key] = (out[0] if isinstance(out, tuple) else out).detach().clone()
        return fn

    # the reference iterates ModuleLists manually (no module forward), so hook the LAST leaf of each block
    handles = []
    for i, blk in enumerate(ref.input_blocks):
        last = blk[-1] if isinstance(blk, torch.nn.ModuleList) else blk
        handles.append(last.register_forward_hook(hook(f"in{i}")))
    handles.append(ref.middle_block[-1].register_forward_hook(hook("mid")))
    for i, blk in enumerate(ref.output_blocks):
        handles.append(blk[-1].register_forward_hook(hook(f"out{i}")))
    with torch.no_grad():
        eps = ref(x, t, y=y, camera_data=cam)
    for h in handles:
        h.remove()
    out = {"x": x, "t": t, "y": y, "camera_data": cam, "eps": eps.contiguous(),
           "weights_checksum": torch.tensor([checksum(sd)], dtype=torch.float64)}
    for k, v in taps.items():
        if v.dim() == 5:  # TemporalTransformer returns b c f h w -> store as (b f) c h w like the others
            b_, c_, f_, h_, w_ = v.shape
            v = v.permute(0, 2, 1, 3, 4).reshape(b_ * f_, c_, h_, w_)
        out[f"tap.{k}"] = v.contiguous()
    save_file(out, os.path.join(GOLD, f"{name}.safetensors"),
              metadata={"cfg": json.dumps(c), "seed": str(seed)})
    print(name, "eps", tuple(eps.shape), float(eps.abs().mean()), "taps", len(taps))
    return ref, sd


def ddim_case(ns, ref, c):
    """BASELINE config-1 analogue: 4 views, 2 DDIM steps, CFG 9, fp32 CPU eager."""
    D = ns.ddim.DiffusionDDIM
    dif = D(schedule="linear_sd", schedule_param=dict(num_timesteps=1000, init_beta=0.00085, last_beta=0.012,
                                                       zero_terminal_snr=False),
            mean_type="eps", loss_type="mse", var_type="fixed_small", rescale_timesteps=False,
            noise_strength=0.0)
    g = torch.Generator().manual_seed(11)
    noise = torch.randn(1, 4, 4, 8, 8, generator=g)
    y = torch.randn(1, 5, c["context_dim"], generator=g)
    y0 = torch.randn(1, 5, c["context_dim"], generator=g)
    cam = torch.randn(1, 4, 16, generator=g)
    kw = [dict(y=y, camera_data=cam), dict(y=y0, camera_data=cam)]
    out = {}
    for n in (2, 5):
        xt = dif.ddim_sample_loop(noise=noise.clone(), model=ref, model_kwargs=kw, guide_scale=9.0,
                                  ddim_timesteps=n, eta=0.0)
        out[f"x0_steps{n}"] = xt.contiguous()
    out.update({"noise": noise, "y": y, "y_uncond": y0, "camera_data": cam})
    save_file(out, os.path.join(GOLD, "ddim_tiny.safetensors"))
    print("ddim", {k: float(v.abs().mean()) for k, v in out.items() if k.startswith("x0")})


def schedule_case(ns):
    D = ns.ddim.DiffusionDDIM
    out = {}
    for tag, kw in (("linear_sd", dict(schedule="linear_sd",
                                       schedule_param=dict(num_timesteps=1000, init_beta=0.00085,
                                                           last_beta=0.012, zero_terminal_snr=False))),
                    ("cosine_ztsnr", dict(schedule="cosine",
                                          schedule_param=dict(num_timesteps=1000, cosine_s=0.008,
                                                              zero_terminal_snr=True)))):
        d = D(mean_type="eps", var_type="fixed_small", **kw)
        for name in ("betas", "alphas_cumprod", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
                     "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod"):
            out[f"{tag}.{name}"] = getattr(d, name).clone()
    for n in (2, 20, 50):
        out[f"steps{n}"] = (1 + torch.arange(0, 1000, 1000 // n)).clamp(0, 999).flip(0)
    save_file(out, os.path.join(GOLD, "schedules.safetensors"))
    print("schedules ac[0], ac[999] =", float(out["linear_sd.alphas_cumprod"][0]),
          float(out["linear_sd.alphas_cumprod"][999]))


def vae_case(ns):
    A = ns.autoencoder.AutoencoderKL
    dd = dict(double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=32,
              ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    vae = A(ddconfig=dd, embed_dim=4).eval()
    shapes = vae_decoder_param_shapes(ch=32)
    ref_sd = {k: v for k, v in vae.state_dict().items()
              if k.startswith("decoder.") or k.startswith("post_quant_conv.")}
    assert set(ref_sd.keys()) == set(shapes.keys()), set(ref_sd.keys()) ^ set(shapes.keys())
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    sd = random_state_dict(shapes, 77)
    vae.load_state_dict(sd, strict=False)
    g = torch.Generator().manual_seed(78)
    z = torch.randn(2, 4, 8, 8, generator=g)
    with torch.no_grad():
        img = vae.decode(z)
    save_file({"z": z, "img": img.contiguous(),
               "weights_checksum": torch.tensor([checksum(sd)], dtype=torch.float64)},
              os.path.join(GOLD, "vae_tiny.safetensors"))
    print("vae", tuple(img.shape), float(img.abs().mean()))


def vae_enc_case(ns):
    from .weights import vae_encoder_param_shapes
    A = ns.autoencoder.AutoencoderKL
    dd = dict(double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=32,
              ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    vae = A(ddconfig=dd, embed_dim=4).eval()
    shapes = vae_encoder_param_shapes(ch=32)
    ref_sd = {k: v for k, v in vae.state_dict().items() if k.startswith("encoder.") or k.startswith("quant_conv.")}
    assert set(ref_sd.keys()) == set(shapes.keys()), set(ref_sd.keys()) ^ set(shapes.keys())
    sd = random_state_dict(shapes, 91)
    vae.load_state_dict(sd, strict=False)
    g = torch.Generator().manual_seed(92)
    img = torch.randn(2, 3, 64, 72, generator=g)
    with torch.no_grad():
        post = vae.encode(img)
    save_file({"img": img, "moments": post.parameters.contiguous(),
               "weights_checksum": torch.tensor([checksum(sd)], dtype=torch.float64)},
              os.path.join(GOLD, "vae_enc_tiny.safetensors"))
    print("vae enc", tuple(post.parameters.shape), float(post.parameters.abs().mean()))


def manifest_case(ns):
    """Full-size key/shape manifests (G7): text, no tensors."""
    full = dict(in_dim=4, dim=320, context_dim=1024, out_dim=4, dim_mult=[1, 2, 4, 4], num_heads=8, head_dim=64,
                num_res_blocks=2, attn_scales=[1.0, 0.5, 0.25])
    with torch.device("meta"):
        ref = build_ref_unet(ns, full)
    man = {k: list(v.shape) for k, v in ref.state_dict().items()}
    n_params = sum(p.numel() for p in ref.parameters())
    with open(os.path.join(GOLD, "manifest_unet_t2v_full.json"), "w") as f:
        json.dump({"n_params": n_params, "n_keys": len(man), "keys": man}, f, indent=0)
    print("manifest", len(man), n_params)
    A = ns.autoencoder.AutoencoderKL
    dd = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
              ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    with torch.device("meta"):
        vae = A(ddconfig=dd, embed_dim=4)
    man = {k: list(v.shape) for k, v in vae.state_dict().items()}
    with open(os.path.join(GOLD, "manifest_vae_full.json"), "w") as f:
        json.dump({"n_keys": len(man), "keys": man}, f, indent=0)


def i2v_case(ns):
    """Tiny UNetSD_I2VGen (I2VGen-XL front-end, BASELINE configs[3] analogue): eps/v + concat + context tokens."""
    import importlib
    from .unet_i2v_ref import i2v_param_shapes
    torch.Tensor.cuda = lambda self, *a, **k: self          # unet_i2vgen.py:334 hard-codes .cuda()
    m = importlib.import_module("tools.modules.unet.unet_i2vgen")
    c = dict(in_dim=4, dim=64, context_dim=1024, out_dim=4, dim_mult=[1, 2], num_heads=2, head_dim=64,
             num_res_blocks=1, attn_scales=[1.0, 0.5])
    ref = m.UNetSD_I2VGen(y_dim=1024, dropout=0.1, temporal_attention=True, use_checkpoint=False,
                          use_camera_condition=True, use_lgm_refine=False, use_fps_condition=False, concat_dim=4,
                          **c).eval()
    cfg = UNetCfg(**c)
    trunk = unet_param_shapes(UNetCfg(**dict(c, in_dim=8)))
    shapes = dict(trunk)
    shapes.update(i2v_param_shapes(cfg))
    ref_sd = ref.state_dict()
    assert set(ref_sd.keys()) == set(shapes.keys()), set(ref_sd.keys()) ^ set(shapes.keys())
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), (k, v.shape, shapes[k])
    ordered = {k: shapes[k] for k in sorted(shapes)}          # generation order = sorted keys (order-independent)
    sd = random_state_dict(ordered, 2024)
    ref.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(2025)
    B, F_, H, W, L = 2, 3, 8, 8, 5
    x = torch.randn(B, 4, F_, H, W, generator=g)
    t = torch.tensor([481, 41])
    y = torch.randn(B, L, 1024, generator=g)
    img = torch.randn(B, 1, 1024, generator=g)
    li = torch.randn(B, 4, H, W, generator=g)
    cam = torch.randn(B, F_, 16, generator=g)
    fps = torch.tensor([8, 8])
    with torch.no_grad():
        out = ref(x, t, y=y, image=img, local_image=li.unsqueeze(2).repeat_interleave(F_, dim=2), fps=fps, camera_data=cam)
    save_file({"x": x, "t": t, "y": y, "image": img, "local_image": li, "camera_data": cam, "fps": fps,
               "out": out.contiguous(), "weights_checksum": torch.tensor([checksum(sd)], dtype=torch.float64)},
              os.path.join(GOLD, "unet_i2v_tiny.safetensors"), metadata={"cfg": json.dumps(c), "seed": "2024"})
    print("i2v", tuple(out.shape), float(out.abs().mean()))


def camera_case():
    """Orbit cameras of the t2v entrance (utils/camera_utils.py get_camera + the row flips at
    inference_text2video_entrance.py:186-191) -> camera_data [1,24,16]."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_camera_utils", os.path.join(shim.REF_ROOT, "utils", "camera_utils.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    cam = m.get_camera(24, elevation=15, azimuth_start=0, azimuth_span=360, camera_distance=2.0).unsqueeze(0)
    cam = cam.reshape(1, 24, 4, 4)
    cam[:, :, 1, :] *= -1
    cam[:, :, [0, 1], :] = cam[:, :, [1, 0], :]
    save_file({"camera_data": cam.reshape(1, 24, 16).contiguous()}, os.path.join(GOLD, "camera_24.safetensors"))


LGM_TINY = dict(down_channels=(64, 512, 1024), down_attention=(False, True, True), mid_attention=True,
                up_channels=(1024, 512, 64), up_attention=(True, True, False))


def lgm_case():
    """LGM branch (core/unet.py, core/models.py forward_gaussians, core/utils.py get_rays) from the imported reference:
    a 3-level U-Net whose attention levels have head_dim 32 and 64 (the two the full model uses), the Gaussian
    activations, the ray helper, and the key manifest of the full 'big' model."""
    import dataclasses
    from .lgm_ref import LgmCfg, lgm_unet_param_shapes
    ns = shim.load_lgm_reference()
    cfg = LgmCfg(**LGM_TINY)
    shapes = lgm_unet_param_shapes(cfg)
    ref = ns.unet.UNet(9, 14, down_channels=cfg.down_channels, down_attention=cfg.down_attention,
                       mid_attention=cfg.mid_attention, up_channels=cfg.up_channels, up_attention=cfg.up_attention).eval()
    ref_sd = ref.state_dict()
    assert list(ref_sd.keys()) == list(shapes.keys()), "LGM U-Net manifest order mismatch"
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), (k, v.shape, shapes[k])
    sd = random_state_dict(shapes, 2468)
    ref.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(97)
    x = torch.randn(4, 9, 32, 32, generator=g)
    taps = {}

    def hook(key):
        def fn(mod, inp, out):
            taps[key] = (out[0] if isinstance(out, tuple) else out).detach().clone()
        return fn
    for i, b in enumerate(ref.down_blocks):
        b.register_forward_hook(hook(f"down_blocks.{i}"))
    ref.mid_block.register_forward_hook(hook("mid_block"))
    for i, b in enumerate(ref.up_blocks):
        b.register_forward_hook(hook(f"up_blocks.{i}"))
    with torch.no_grad():
        out = ref(x)
    tensors = {"x": x, "out": out.contiguous()}
    for k in ("down_blocks.2", "mid_block"):
        tensors["tap." + k] = taps[k].contiguous()
    save_file(tensors, os.path.join(GOLD, "lgm_unet_tiny.safetensors"),
              metadata={"cfg": json.dumps(LGM_TINY), "seed": "2468", "weights_checksum": str(checksum(sd))})
    # forward_gaussians (U-Net + 1x1 conv + activations) through the reference LGM class
    opt = dataclasses.replace(ns.options.config_defaults["big"], down_channels=cfg.down_channels,
                              down_attention=cfg.down_attention, mid_attention=cfg.mid_attention,
                              up_channels=cfg.up_channels, up_attention=cfg.up_attention, input_size=32, splat_size=32,
                              output_size=64, lambda_lpips=0.0)
    lgm = ns.models.LGM(opt).eval()
    lsd = {("unet." + k): v for k, v in sd.items()}
    gc = torch.Generator().manual_seed(5)
    lsd["conv.weight"] = (torch.randn(14, 14, 1, 1, generator=gc) * 0.3).bfloat16().float()
    lsd["conv.bias"] = (torch.randn(14, generator=gc) * 0.1).bfloat16().float()
    assert set(lgm.state_dict().keys()) == set(lsd.keys())
    lgm.load_state_dict(lsd, strict=True)
    with torch.no_grad():
        gauss = lgm.forward_gaussians(x.unsqueeze(0))
    save_file({"images": x.unsqueeze(0).contiguous(), "gaussians": gauss.contiguous(), "conv.weight": lsd["conv.weight"],
               "conv.bias": lsd["conv.bias"]}, os.path.join(GOLD, "lgm_gaussians_tiny.safetensors"),
              metadata={"cfg": json.dumps(LGM_TINY), "seed": "2468"})
    # rays of two poses
    poses = torch.eye(4).repeat(2, 1, 1)
    poses[0, :3, 3] = torch.tensor([0.0, 0.0, 1.5])
    c, s_ = 0.8, 0.6
    poses[1, :3, :3] = torch.tensor([[c, 0.0, s_], [0.0, 1.0, 0.0], [-s_, 0.0, c]])
    poses[1, :3, 3] = torch.tensor([0.9, 0.2, 1.2])
    ro, rd = zip(*[ns.utils.get_rays(poses[i], 8, 12, 39.6) for i in range(2)])
    save_file({"poses": poses, "rays_o": torch.stack(ro).contiguous(), "rays_d": torch.stack(rd).contiguous()},
              os.path.join(GOLD, "lgm_rays.safetensors"))
    # key manifest of the full model as the UNet registers it (unet_t2v.py:267-274: self.lgm_big = LGM(opt))
    big = ns.models.LGM(dataclasses.replace(ns.options.config_defaults["big"], lambda_lpips=0.0))
    man = {k: list(v.shape) for k, v in big.state_dict().items()}
    with open(os.path.join(GOLD, "manifest_lgm_big.json"), "w") as f:
        json.dump({"n_keys": len(man), "n_params": int(sum(v.numel() for v in big.state_dict().values())), "shapes": man}, f)
    print("lgm: golden written;", len(man), "keys,", sum(v.numel() for v in big.state_dict().values()), "params")


def main():
    os.makedirs(GOLD, exist_ok=True)
    ns = shim.load_reference()
    torch.manual_seed(0)
    schedule_case(ns)
    ref_a, _ = unet_case(ns, "unet_tiny_a", TINY_A, seed=1234, F_=4, H=8, W=8, L=7)
    ddim_case(ns, ref_a, TINY_A)
    unet_case(ns, "unet_tiny_b", TINY_B, seed=4321, F_=3, H=8, W=12, L=5)
    vae_case(ns)
    vae_enc_case(ns)
    manifest_case(ns)
    camera_case()
    i2v_case(ns)
    lgm_case()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""End-to-end effect of videomv_amd/tuned_gemm.json on RESULTS: the same seeded inputs through (a) the table and (b) the built-in
policy — a plain step and an LGM-refined step at 24x32x32, and the 24-frame VAE decode at 320x512.  Other tiles / split-K factors round
differently (expected rel-L2 ~1e-3, the storage-rounding noise); a wrong kernel choice would be O(1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from videomv_amd import ops
from videomv_amd.registry import MODEL, DIFFUSION, AUTO_ENCODER
import videomv_amd.unet_t2v, videomv_amd.diffusion_ddim, videomv_amd.autoencoder  # noqa
from videomv_amd.lgm import prepare_gs_data
from videomv_amd.camera import entrance_camera_data
from videomv_amd.pipeline import decode_views

dev = torch.device("cuda", 0)
dd = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)
with torch.device(dev):
    vae = AUTO_ENCODER.build(dict(type="AutoencoderKL", ddconfig=dd, embed_dim=4))
    m = MODEL.build(dict(type="UNetSD_T2VBase", y_dim=1024, use_camera_condition=True, use_lgm_refine=True, **bench.FULL))
bench.randomize_(vae, 4321); bench.randomize_(m, 1234); m.eval()
dif = DIFFUSION.build(dict(type="DiffusionDDIM", schedule="linear_sd", schedule_param=dict(num_timesteps=1000, init_beta=0.00085, last_beta=0.012, zero_terminal_snr=False), mean_type="eps", var_type="fixed_small"))
g = torch.Generator(device=dev).manual_seed(11)
x32 = torch.randn(1, 4, 24, 32, 32, generator=g, device=dev)
x64 = torch.randn(1, 4, 24, 40, 64, generator=g, device=dev)
y, y0 = torch.randn(1, 77, 1024, generator=g, device=dev), torch.randn(1, 77, 1024, generator=g, device=dev)
cam = entrance_camera_data(24, elevation=15, camera_distance=2.0)
gs = prepare_gs_data(cam, m.lgm_opt)
kw = [dict(y=y, camera_data=cam, gs_data=gs), dict(y=y0, camera_data=cam, gs_data=gs)]
res = {}
for tag in ("tuned", "policy"):
    if tag == "policy":
        ops._TUNED = {}
        m._invalidate(); vae._engines.clear()
    a = x32.clone(); dif.ddim_step_hip(a, 501, m, kw[0], kw[1], 9.0, 20)
    b = x32.clone(); torch.manual_seed(9); dif.ddim_step_lgm(b, 581, m, kw[0], kw[1], 9.0, 20, vae)
    c = x64.clone(); dif.ddim_step_hip(c, 501, m, dict(y=y, camera_data=cam), dict(y=y0, camera_data=cam), 9.0, 20)
    v = decode_views(vae, x64 * 0.18215)
    torch.cuda.synchronize()
    res[tag] = (a, b, c, v)
    print(tag, "tuned GEMMs in UNet plans:", sum(e.n_tuned for e in m._engines.values()), "VAE:", sum(getattr(e, "n_tuned", 0) for e in vae._engines.values()))
rl = lambda p, q: float((p.float() - q.float()).norm() / q.float().norm())
for name, i in (("plain step 24x32x32", 0), ("LGM-refined step 24x32x32", 1), ("plain step 24x40x64", 2), ("VAE decode 24 x 320x512", 3)):
    print(f"{name:28s} tuned vs policy rel-L2 {rl(res['tuned'][i], res['policy'][i]):.3e}  finite {bool(torch.isfinite(res['tuned'][i]).all())}")

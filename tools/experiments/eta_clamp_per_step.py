#!/usr/bin/env python
"""GPU: per-step (teacher-forced) error of the fused CFG + DDIM update with the sampler options eta / clamp against the oracle loop.
Why it exists: the free-running 4-step CFG-9 loop with a hard clamp amplifies the per-step 16-bit error to ~4e-2, which first looked like a
formula mismatch; fed the oracle's x_t at every step the HIP step agrees to 3e-4 .. 7e-3 (tests/test_unet_gpu.py::
test_fused_step_with_clamp_and_eta_matches_oracle asserts exactly this)."""
import sys, os, torch, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.unet_ref import UNetCfg, unet_forward
from oracle.weights import random_state_dict, unet_param_shapes
from oracle.ddim_ref import betas_for, DDIMTables, ddim_sample_loop
from videomv_amd.registry import DIFFUSION, MODEL
import videomv_amd
cfg = dict(in_dim=4, dim=64, context_dim=1024, out_dim=4, dim_mult=[1, 2], num_heads=2, head_dim=64, num_res_blocks=1, attn_scales=[1.0, 0.5])
ocfg = UNetCfg(**cfg)
sd = random_state_dict(unet_param_shapes(ocfg), 31)
m = MODEL.build(dict(type="UNetSD_T2VBase", in_dim=4, dim=64, y_dim=1024, context_dim=1024, out_dim=4, dim_mult=[1, 2], num_heads=2, head_dim=64,
                     num_res_blocks=1, attn_scales=[1.0, 0.5], use_camera_condition=True, use_lgm_refine=False))
m.load_state_dict(sd, strict=True); m = m.eval().cuda()
dif = DIFFUSION.build(dict(type="DiffusionDDIM", schedule="linear_sd", schedule_param=dict(num_timesteps=1000, init_beta=0.00085, last_beta=0.012, zero_terminal_snr=False),
                           mean_type="eps", var_type="fixed_small"))
gen = torch.Generator().manual_seed(12)
noise = torch.randn(1, 4, 4, 8, 8, generator=gen)
y, y0 = torch.randn(1, 7, 1024, generator=gen), torch.randn(1, 7, 1024, generator=gen)
cam = torch.randn(1, 4, 16, generator=gen)
steps = dif.ddim_steps(3); n = len(steps)
nz = [torch.randn(1, 4, 4, 8, 8, generator=gen) for _ in range(n)]
kw = [dict(y=y.cuda(), camera_data=cam), dict(y=y0.cuda(), camera_data=cam)]
tb = DDIMTables(betas_for("linear_sd"))
rel = lambda a, b: float((a.float().cpu() - b.float().cpu()).norm() / b.float().cpu().norm())
for eta, clamp in ((0.0, None), (0.6, None), (0.0, 2.5), (0.6, 2.5)):
    trace = []
    ddim_sample_loop(noise.clone(), lambda xt, t, y, camera_data: unet_forward(sd, ocfg, xt, t, y, camera_data), tb,
                     [dict(y=y, camera_data=cam), dict(y=y0, camera_data=cam)], 9.0, ddim_timesteps=3, eta=eta, clamp=clamp, trace=trace, step_noise=lambda i, xt: nz[i])
    xt = noise.clone().cuda().float().contiguous()
    m.begin_sample()
    out = []
    for i, step in enumerate(steps):
        # teacher-forced: feed the ORACLE's x_t of this step, so each line is one step's own error
        xin = (noise if i == 0 else trace[i - 1]).clone().cuda().float().contiguous()
        real = torch.randn_like
        torch.randn_like = lambda t, *a, **k: nz[i].to(t.device)
        dif.ddim_step_hip(xin, int(step), m, kw[0], kw[1], 9.0, 1000 // 3, clamp=clamp, eta=eta)
        torch.randn_like = real
        out.append(round(rel(xin, trace[i]), 5))
    print("eta", eta, "clamp", clamp, "per-step rel-L2 (teacher-forced):", out, "sigma", [round(dif.ddim_sigma(int(s), 333, eta), 4) for s in steps])

"""Fused denoising step with b = 1 .. 4 prompts per plan (unet_t2v._forward_cfg_rows_batched) at the bench shape and at the reference's
own: ms per batched step and sample-steps/s.  python tools/experiments/prompt_batch_scaling.py [40x64] [bmax]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from videomv_amd.registry import MODEL, DIFFUSION
import videomv_amd  # noqa: F401
import videomv_amd.diffusion_ddim  # noqa: F401
from videomv_amd.camera import entrance_camera_data

H, W = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "40x64").split("x"))
bmax = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
with torch.device(dev):
    m = MODEL.build(dict(type="UNetSD_T2VBase", y_dim=1024, use_camera_condition=True, use_lgm_refine=False, **bench.FULL))
bench.randomize_(m, 1234)
m.eval()
dif = DIFFUSION.build(dict(type="DiffusionDDIM", schedule="linear_sd", schedule_param=dict(num_timesteps=1000, init_beta=0.00085, last_beta=0.012,
                                                                                              zero_terminal_snr=False), mean_type="eps", var_type="fixed_small"))
g = torch.Generator(device=dev).manual_seed(11)
cam = entrance_camera_data(24, elevation=15, camera_distance=2.0).to(dev)
steps = [int(s) for s in dif.ddim_steps(50)]
y0 = torch.randn(1, 77, 1024, generator=g, device=dev)
base = None
for b in list(range(1, bmax + 1)) + [1]:
    x = torch.randn(b, 4, 24, H, W, generator=g, device=dev)
    kc, ku = dict(y=torch.randn(b, 77, 1024, generator=g, device=dev), camera_data=cam), dict(y=y0, camera_data=cam)
    for i in range(2):
        dif.ddim_step_hip(x, steps[i], m, kc, ku, 9.0, 20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 8
    for i in range(n):
        dif.ddim_step_hip(x, steps[2 + i], m, kc, ku, 9.0, 20)
    torch.cuda.synchronize()
    ms = 1000 * (time.perf_counter() - t0) / n
    base = base or ms
    tf = bench.STEP_TFLOP.get((H, W))
    print(f"24x{H}x{W}  prompts/plan {b}: {ms:7.2f} ms per batched step = {ms / b:6.2f} ms per sample-step, {1000 * b / ms:6.2f} sample-steps/s, "
          f"x{b * base / ms:.3f} vs 1 prompt" + (f", {b * tf / (ms * 1e-3) / 2500:.3f} of MFMA peak" if tf else "") + f", finite={bool(torch.isfinite(x).all())}", flush=True)
    for k_ in [k for k in m._engines if k[0] > 2]:
        m._engines.pop(k_, None)
    torch.cuda.empty_cache()

"""Generate ``tests/golden/ddim50_full_24x16x16.safetensors`` from the IMPORTED reference (authoring container only;
~15-25 min of CPU, which is why it is not part of ``make_golden.py``).

    python -m oracle.make_golden_ddim50 [--threads N]

SURVEY §8d asks for the 50-step figure against the fp32 reference.  The full 1.413 B ``UNetSD_T2VBase`` at the bench
shape (24 x 40 x 64) costs ~50 s per forward on the host, i.e. 100 forwards are out of reach; the SAME architecture at
24 x 16 x 16 (every level still has whole tiles: 16^2 / 8^2 / 4^2 / 2^2 pixels) runs the reference's own
``DiffusionDDIM.ddim_sample_loop`` (50 steps, CFG 9, eta 0, linear_sd schedule: t2v_infer.yaml) in that time.  The fixture
stores the inputs and the reference's final latent; ``tests/test_unet_gpu.py::test_ddim50_full_arch_vs_reference_fixture``
runs the HIP loop on the same inputs with the same seeded weights (oracle/weights.py) and reports PSNR + rel-L2.
The fixture is data (tensors); no reference source is copied.
"""
import argparse
import json
import os
import time

import torch
from safetensors.torch import save_file

from . import shim
from .weights import random_state_dict, unet_param_shapes, checksum
from .unet_ref import UNetCfg
from .make_golden import build_ref_unet, GOLD

FULL = dict(in_dim=4, dim=320, context_dim=1024, out_dim=4, dim_mult=[1, 2, 4, 4], num_heads=8, head_dim=64,
            num_res_blocks=2, attn_scales=[1.0, 0.5, 0.25])
SEED_W, SEED_X, F_, H, W, STEPS, GUIDE = 5, 11, 24, 16, 16, 50, 9.0


def orbit_cameras(frames):
    """The entrance's cameras (videomv_amd/camera.py is this repo's own code, pinned to the reference by
    tests/golden/camera_24.safetensors)."""
    from videomv_amd.camera import entrance_camera_data
    return entrance_camera_data(frames, elevation=15, camera_distance=2.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=max(1, (os.cpu_count() or 2) - 1))
    ap.add_argument("--steps", type=int, default=STEPS)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    ns = shim.load_reference()
    cfg = UNetCfg(**FULL)
    shapes = unet_param_shapes(cfg)
    ref = build_ref_unet(ns, FULL)
    sd = random_state_dict(shapes, SEED_W)
    ref.load_state_dict(sd, strict=True)
    D = ns.ddim.DiffusionDDIM
    dif = D(schedule="linear_sd", schedule_param=dict(num_timesteps=1000, init_beta=0.00085, last_beta=0.012,
                                                       zero_terminal_snr=False),
            mean_type="eps", loss_type="mse", var_type="fixed_small", rescale_timesteps=False, noise_strength=0.0)
    g = torch.Generator().manual_seed(SEED_X)
    noise = torch.randn(1, 4, F_, H, W, generator=g)
    y = torch.randn(1, 77, 1024, generator=g)
    y0 = torch.randn(1, 77, 1024, generator=g)
    cam = orbit_cameras(F_)
    kw = [dict(y=y, camera_data=cam), dict(y=y0, camera_data=cam)]
    t0 = time.time()
    torch.manual_seed(0)          # (the reference draws an unused randn_like per step: diffusion_ddim.py:240-243)
    with torch.no_grad():
        x0 = dif.ddim_sample_loop(noise=noise.clone(), model=ref, model_kwargs=kw, guide_scale=GUIDE,
                                  ddim_timesteps=a.steps, eta=0.0)
    dt = time.time() - t0
    name = f"ddim{a.steps}_full_{F_}x{H}x{W}.safetensors"
    save_file({"noise": noise, "y": y, "y_uncond": y0, "camera_data": cam.contiguous(), "x0": x0.contiguous(),
               "weights_checksum": torch.tensor([checksum(sd)], dtype=torch.float64)},
              os.path.join(GOLD, name),
              metadata={"cfg": json.dumps(FULL), "seed": str(SEED_W), "steps": str(a.steps), "guide_scale": str(GUIDE),
                        "source": "imported reference: UNetSD_T2VBase + DiffusionDDIM.ddim_sample_loop, fp32 CPU eager"})
    print(name, "x0 mean|.|", float(x0.abs().mean()), "std", float(x0.std()), f"{dt:.0f} s on {a.threads} threads")


if __name__ == "__main__":
    main()

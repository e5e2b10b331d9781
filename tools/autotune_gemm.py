#!/usr/bin/env python
"""Measure, per distinct implicit-GEMM launch of the recorded UNet plans, every kernel configuration the library accepts for it —
tile family x split-K factor — on the GPU at hand, and write the choices that beat the built-in policy to
videomv_amd/tuned_gemm.json (read by ops.tuned_table(); UNetEngine._tune applies it when a plan is recorded).

    python tools/autotune_gemm.py [--worlds 1,8] [--latent 40x64] [--out videomv_amd/tuned_gemm.json] [--merge]

Why: csrc/gemm.hip's policy was fitted by hand to the 1-GPU shapes (M = 122 880 / 30 720 / 7 680 / 1 920).  A frame-parallel rank of
an 8-GPU run (BASELINE configs[2]) sees M / 8 — 15 360 / 3 840 / 960 / 240 rows against the same weights — where the first
simulated-rank measurement of round 4 found the GEMM family at 300 TFLOP/s (a 20-ms rank step for 1/8 of a 50-ms step).
Plans tuned: world 1 = the batched [cond | uncond] plan with the shared CFG prefix; world W > 1 = rank 0's plans (B = 2 single-plan
and B = 1 branch-pipelined) with the peers simulated (comm.SimComm), whose GEMM shapes are the real rank's.

Method per signature (ops.gemm_signature): the recorded argument block is replayed eagerly on its own buffers; a candidate is kept
only if the library accepts it (return code 0), its output agrees with the policy's (rel-L2 <= 2e-3: other tiles / split-K round
differently, a wrong kernel does not), and it is >= 7 % and >= 1.5 us faster in TWO separate timings.  Burst timings of the big
kernels overstate denser kernels on this part (DVFS), so signatures whose policy time is >= 120 us need >= 10 %.
The table is data, not code: a stale entry is refused by vmv_gemm at record time (forced tiles are validated), never silent.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

os.environ["VMV_TUNED"] = "0"            # measure against the built-in policy, not against an older table
from videomv_amd import _lib as L, ops   # noqa: E402

FULL = dict(in_dim=4, dim=320, context_dim=1024, out_dim=4, dim_mult=[1, 2, 4, 4], num_heads=8, head_dim=64,
            num_res_blocks=2, attn_scales=[1.0, 0.5, 0.25])
from videomv_amd.autotune import tune_plan, WS_CAP      # noqa: E402  (the measurement itself lives in the package)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", default="1,8")
    ap.add_argument("--latent", default="40x64")
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--out", default=os.path.join(ROOT, "videomv_amd", "tuned_gemm.json"))
    ap.add_argument("--merge", action="store_true", help="keep the entries already in --out for signatures not measured now")
    ap.add_argument("--min-gain", type=float, default=0.93, help="keep a candidate only if its time <= this fraction of the policy's (launches < 120 us)")
    ap.add_argument("--min-gain-big", type=float, default=0.90, help="the same for launches >= 120 us (burst timings overstate dense kernels)")
    ap.add_argument("--prompts", type=int, default=1, help="world 1: tune the plan of this many prompts per step (B = 2 x prompts row blocks, "
                    "unet_t2v._forward_cfg_rows_batched — the entrance's prompt_batch)")
    ap.add_argument("--lgm", action="store_true", help="also tune the VAE decoder / encoder and LGM U-Net plans (one LGM-refined step at 24x32x32 + the 24-frame decode)")
    a = ap.parse_args()
    H, W = (int(v) for v in a.latent.split("x"))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    import bench
    from videomv_amd.registry import MODEL
    import videomv_amd.unet_t2v  # noqa: F401
    from videomv_amd.comm import SimComm
    from videomv_amd.camera import entrance_camera_data
    with torch.device(dev):
        model = MODEL.build(dict(type="UNetSD_T2VBase", y_dim=1024, use_camera_condition=True, use_lgm_refine=False, **FULL))
    bench.randomize_(model, 1234)
    model.eval()
    g = torch.Generator(device=dev).manual_seed(11)
    noise = torch.randn(1, 4, a.frames, H, W, generator=g, device=dev)
    y, y0 = torch.randn(1, 77, 1024, generator=g, device=dev), torch.randn(1, 77, 1024, generator=g, device=dev)
    cam = entrance_camera_data(a.frames, elevation=15, camera_distance=2.0).to(dev)
    kc, ku = dict(y=y, camera_data=cam), dict(y=y0, camera_data=cam)
    t = torch.tensor([501], device=dev)
    ws = torch.empty(WS_CAP, dtype=torch.uint8, device=dev)
    table = dict(done=set(), entries={})
    t0 = time.time()
    old = {}
    if a.merge and os.path.exists(a.out):
        with open(a.out) as f:
            old = json.load(f)
    try:
        commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    except Exception:
        commit = "?"

    def save():
        merged = dict(old.get(L.elem_name(), {}))
        merged.update(table["entries"])
        other = "bf16" if L.elem_name() == "fp16" else "fp16"
        out = dict(meta=dict(tool="tools/autotune_gemm.py", device=torch.cuda.get_device_name(0), dtype_measured=L.elem_name(), commit=commit,
                             worlds=a.worlds, latent=a.latent, signatures_measured=len(table["done"]), seconds=round(time.time() - t0, 1),
                             rule=">= 7 % (>= 10 % for >= 120-us launches) and >= 1.5 us faster than the policy in two timings; output within 2e-3 rel-L2"))
        out[L.elem_name()] = merged
        if old.get(other):                        # (a section measured with the other element type's library is kept; without one that
            out[other] = dict(old[other])         #  library falls back to this table: same kernels, same shapes — ops.tuned_table)
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)

    for w in [int(v) for v in a.worlds.split(",") if v.strip()]:
        if w < 1:
            continue
        if w == 1:
            model.set_frame_parallel(None)
            if a.prompts > 1:
                nz = torch.randn(a.prompts, 4, a.frames, H, W, generator=g, device=dev)
                yb = torch.randn(a.prompts, 77, 1024, generator=g, device=dev)
                eng, _ = model.forward_cfg_rows(nz, t.expand(a.prompts), dict(y=yb, camera_data=cam), ku)
            else:
                eng, _ = model.forward_cfg_rows(noise, t, kc, ku)
            tune_plan(eng, table, ws, f"world1 {H}x{W}" + (f" prompts{a.prompts}" if a.prompts > 1 else ""), (a.min_gain, a.min_gain_big))
            save()
            continue
        if a.frames % w:
            continue
        xs = noise[:, :, : a.frames // w].contiguous()
        for pipe in ("0", "1"):
            os.environ["VMV_FP_PIPELINE"] = pipe
            model.set_frame_parallel(SimComm(w, 0))
            eng, _ = model.forward_cfg_rows(xs, t, kc, ku)
            e = model._pipe["engs"][0] if pipe == "1" else eng
            tune_plan(e, table, ws, f"world{w} rank0 B={'1' if pipe == '1' else '2'} {H}x{W}", (a.min_gain, a.min_gain_big))
            save()
        os.environ.pop("VMV_FP_PIPELINE", None)
    model.set_frame_parallel(None)
    save()
    if a.lgm:
        # BASELINE configs[4] at its own shape: one LGM-refined step builds every other engine of the sampler — the VAE decoder (8 views
        # at 256 px) and encoder (48 views), the LGM U-Net plans — and the 24-frame decode of the headline shape; their GEMMs get the same
        # treatment (engines take the table through ops.make_tuner)
        from videomv_amd.registry import AUTO_ENCODER, DIFFUSION
        import videomv_amd.autoencoder  # noqa: F401
        import videomv_amd.diffusion_ddim  # noqa: F401
        from videomv_amd.lgm import prepare_gs_data
        from videomv_amd.pipeline import decode_views
        dd = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
                  num_res_blocks=2, attn_resolutions=[], dropout=0.0)
        with torch.device(dev):
            vae = AUTO_ENCODER.build(dict(type="AutoencoderKL", ddconfig=dd, embed_dim=4))
            model_l = MODEL.build(dict(type="UNetSD_T2VBase", y_dim=1024, use_camera_condition=True, use_lgm_refine=True, **FULL))
        bench.randomize_(vae, 4321)
        bench.randomize_(model_l, 1234)
        model_l.eval()
        dif = DIFFUSION.build(dict(type="DiffusionDDIM", schedule="linear_sd",
                                   schedule_param=dict(num_timesteps=1000, init_beta=0.00085, last_beta=0.012, zero_terminal_snr=False),
                                   mean_type="eps", var_type="fixed_small"))
        cam_l = entrance_camera_data(24, elevation=15, camera_distance=2.0)
        gs_data = prepare_gs_data(cam_l, model_l.lgm_opt)
        xl = torch.randn(1, 4, 24, 32, 32, generator=g, device=dev)
        kl = [dict(y=y, camera_data=cam_l, gs_data=gs_data), dict(y=y0, camera_data=cam_l, gs_data=gs_data)]
        dif.ddim_step_lgm(xl, 501, model_l, kl[0], kl[1], 9.0, 20, vae)
        decode_views(vae, noise)
        torch.cuda.synchronize()
        ref = model_l.lgm_refiner(dev)
        engines = [(f"vae {k}", e) for k, e in vae._engines.items()] + [("lgm b1", ref.engine)] + ([("lgm b2", ref._engine2)] if ref._engine2 is not None else [])
        for tag, e in engines:
            tune_plan(e, table, ws, tag if isinstance(tag, str) else str(tag), (a.min_gain, a.min_gain_big))
            save()
    gain = sum(e["base_us"] - e["us"] for e in table["entries"].values())
    print(f"{len(table['entries'])} of {len(table['done'])} signatures improved; sum of per-signature gains {gain:.0f} us; {time.time() - t0:.0f} s -> {a.out}")


if __name__ == "__main__":
    main()

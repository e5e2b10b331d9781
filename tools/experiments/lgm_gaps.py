#!/usr/bin/env python
"""Where an LGM-refined step's wall time goes (GPU only).
    run:      rocprofv3 --kernel-trace -f csv -d OUT -- python tools/experiments/lgm_gaps.py run
    analyse:  python tools/experiments/lgm_gaps.py gaps OUT      (GPU-busy time and the idle gaps > 30 us by the kernel that ends them,
              over the timed calls: the trace window after the marker launch, an fp64 fill)"""
import csv, glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def run():
    import torch
    import bench
    from videomv_amd.registry import MODEL, AUTO_ENCODER, DIFFUSION
    import videomv_amd  # noqa: F401
    from videomv_amd.lgm import prepare_gs_data
    from videomv_amd.camera import entrance_camera_data
    dev = torch.device("cuda", 0)
    with torch.device(dev):
        m = MODEL.build(dict(type="UNetSD_T2VBase", y_dim=1024, use_camera_condition=True, use_lgm_refine=True, **bench.FULL))
        vae = AUTO_ENCODER.build(dict(type="AutoencoderKL", embed_dim=4, ddconfig=dict(double_z=True, z_channels=4, resolution=256,
                                 in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)))
    bench.randomize_(m, 1234); bench.randomize_(vae, 4321)
    m.eval()
    dif = DIFFUSION.build(dict(type="DiffusionDDIM", schedule="linear_sd", schedule_param=dict(num_timesteps=1000, init_beta=0.00085, last_beta=0.012,
                               zero_terminal_snr=False), mean_type="eps", var_type="fixed_small"))
    cam = entrance_camera_data(24, elevation=15, camera_distance=2.0)
    gs = prepare_gs_data(cam, m.lgm_opt)
    g = torch.Generator(device=dev).manual_seed(11)
    y = torch.randn(1, 77, 1024, generator=g, device=dev); y0 = torch.randn(1, 77, 1024, generator=g, device=dev)
    x = torch.randn(1, 4, 24, 32, 32, generator=g, device=dev)
    kw = [dict(y=y, camera_data=cam, gs_data=gs), dict(y=y0, camera_data=cam, gs_data=gs)]
    stride = 1000 // 50
    steps = [int(v) for v in dif.ddim_steps(50)]
    dif.ddim_step_hip(x, steps[0], m, kw[0], kw[1], 9.0, stride)
    for i in range(3):
        dif.ddim_step_lgm(x, steps[1 + i], m, kw[0], kw[1], 9.0, stride, vae)
    torch.cuda.synchronize()
    torch.full((12345,), 1.0, device=dev, dtype=torch.float64)      # marker launch (the only fp64 fill): the analysis starts after it
    torch.cuda.synchronize()
    ts = []
    for i in range(5):
        t0 = time.perf_counter()
        dif.ddim_step_lgm(x, steps[4 + i], m, kw[0], kw[1], 9.0, stride, vae)
        torch.cuda.synchronize()
        ts.append(1000 * (time.perf_counter() - t0))
    print("lgm-refined step wall ms:", [round(t, 2) for t in ts])


def gaps(d):
    f = [p for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)][0]
    rows = list(csv.DictReader(open(f)))
    ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
    idx = max(i for i, k in enumerate(ks) if "FillFunctor<double>" in k[2])
    ks = ks[idx + 1:]
    t0, t1 = ks[0][0], max(k[1] for k in ks)
    busy, end, by, prev = 0, ks[0][0], {}, ""
    for s, e, n in ks:
        if s > end:
            gap = s - end
            if gap > 1000000:
                print(f"  gap {1e-6 * gap:7.3f} ms at +{1e-6 * (end - t0):8.2f} ms  after {prev[:60]}  before {n[:60]}")
            if gap > 30000:
                key = n[:70]
                by[key] = by.get(key, [0, 0]); by[key][0] += 1; by[key][1] += gap
            busy += e - s
        else:
            busy += max(0, e - max(s, end))
        if e >= end:
            prev = n
        end = max(end, e)
    print(f"window {1e-6 * (t1 - t0):.2f} ms over 5 steps, GPU busy {1e-6 * busy:.2f} ms ({100.0 * busy / (t1 - t0):.1f} %), kernels {len(ks)}")
    for k, (c, g) in sorted(by.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"  idle {1e-6 * g:8.3f} ms in {c:4d} gaps before  {k}")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else gaps(sys.argv[2])

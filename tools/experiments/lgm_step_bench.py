"""Times the stages of one LGM-refined branch at full size (GPU only): VAE decode of 4 views, LGM U-Net, 24 renders, VAE
encode of 24 views.  Used with rocprofv3 --kernel-trace --stats for profiles/r1_lgm_kernel_stats.txt."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from videomv_amd.registry import MODEL, AUTO_ENCODER
import videomv_amd
from videomv_amd.lgm import prepare_gs_data
from videomv_amd.camera import entrance_camera_data

dev = torch.device("cuda", 0)
with torch.device(dev):
    m = MODEL.build(dict(type="UNetSD_T2VBase", y_dim=1024, use_camera_condition=True, use_lgm_refine=True, **bench.FULL))
    vae = AUTO_ENCODER.build(dict(type="AutoencoderKL", embed_dim=4, ddconfig=dict(double_z=True, z_channels=4, resolution=256,
                             in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)))
bench.randomize_(m, 1234); bench.randomize_(vae, 4321)
ref = m.lgm_refiner(dev)
cam = entrance_camera_data(24, elevation=15, camera_distance=2.0)
gs = prepare_gs_data(cam, m.lgm_opt)
g = torch.Generator(device=dev).manual_seed(1)
xt = torch.randn(1, 4, 24, 32, 32, generator=g, device=dev)
eps = torch.randn(2 * 24 * 1024, 4, generator=g, device=dev)
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for rep in range(3):
    t0 = sync()
    ref.latent_z(eps, 4, 0, xt, 1.1, 0.5, vae, gs)
    t1 = sync()
print("latent_z total ms", 1000 * (t1 - t0))
for rep in range(3):
    t0 = sync()
    ref.latent_z_pair(eps, 4, xt, 1.1, 0.5, vae, gs)
    t1 = sync()
print("latent_z_pair (both CFG branches batched) total ms", 1000 * (t1 - t0), "= per branch", 500 * (t1 - t0))
# stages
z4 = torch.randn(4, 4, 32, 32, generator=g, device=dev)
t0 = sync(); dec = vae.decode(z4); t1 = sync()
inp = torch.randn(4, 9, 256, 256, generator=g, device=dev)
ga = ref.engine.forward_gaussians(inp); t2 = sync()
out = ref.renderer.render(ga.unsqueeze(0), gs["cam_view"].to(dev), gs["cam_view_proj"].to(dev), None, bg_color=torch.full((3,), 0.5, device=dev)); t3 = sync()
small = torch.rand(24, 3, 256, 256, generator=g, device=dev) * 2 - 1
z = vae.encode_firsr_stage(small, 0.18215); t4 = sync()
print(f"decode4 {1000*(t1-t0):.1f} ms | lgm unet {1000*(t2-t1):.1f} ms | 24 renders {1000*(t3-t2):.1f} ms ({sum(ref.renderer.last_num_rendered)/24/1e6:.2f} M inst/view) | encode24 {1000*(t4-t3):.1f} ms")


# ---- per-op tables of the three recorded plans (events on the launch stream, min of 3 replays)
from videomv_amd import _lib as L
from videomv_amd.flops import gemm_flops, attn_flops


def op_table(S, name):
    n = S.nops
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(n + 1)] for _ in range(3)]
    for r in range(3):
        torch.cuda.synchronize()
        ev[r][0].record()
        for i in range(n):
            S.run(i, i + 1)
            ev[r][i + 1].record()
    torch.cuda.synchronize()
    ms = [min(ev[r][i].elapsed_time(ev[r][i + 1]) for r in range(3)) for i in range(n)]
    path = os.path.join(os.environ.get("VMV_OUT", "gpurun_out"), f"r2_ops_{name}.tsv")
    with open(path, "w") as f:
        f.write("idx\tlabel\tms\tGFLOP\tTFLOP/s\tM\tN\tK\n")
        for i, (op, p) in enumerate(S.recorded):
            fl = gemm_flops(p) if op == L.OP_GEMM else (attn_flops(p) if op == L.OP_ATTENTION else 0.0)
            mnk = (p.M, p.N, p.ktot) if op == L.OP_GEMM else ("", "", "")
            f.write(f"{i}\t{S.labels[i]}\t{ms[i]:.4f}\t{fl / 1e9:.2f}\t{(fl / (ms[i] * 1e-3) / 1e12) if fl else 0:.1f}\t{mnk[0]}\t{mnk[1]}\t{mnk[2]}\n")
    print(name, "ops", n, "sum ms", round(sum(ms), 2))


op_table(ref.engine.S, "lgm_unet")
for key, e in vae._engines.items():
    op_table(e.S, f"vae_{type(e).__name__}_{key[0]}x{key[1]}x{key[2]}")

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
# stages
z4 = torch.randn(4, 4, 32, 32, generator=g, device=dev)
t0 = sync(); dec = vae.decode(z4); t1 = sync()
inp = torch.randn(4, 9, 256, 256, generator=g, device=dev)
ga = ref.engine.forward_gaussians(inp); t2 = sync()
out = ref.renderer.render(ga.unsqueeze(0), gs["cam_view"].to(dev), gs["cam_view_proj"].to(dev), None, bg_color=torch.full((3,), 0.5, device=dev)); t3 = sync()
small = torch.rand(24, 3, 256, 256, generator=g, device=dev) * 2 - 1
z = vae.encode_firsr_stage(small, 0.18215); t4 = sync()
print(f"decode4 {1000*(t1-t0):.1f} ms | lgm unet {1000*(t2-t1):.1f} ms | 24 renders {1000*(t3-t2):.1f} ms ({sum(ref.renderer.last_num_rendered)/24/1e6:.2f} M inst/view) | encode24 {1000*(t4-t3):.1f} ms")

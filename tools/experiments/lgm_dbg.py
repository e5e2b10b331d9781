"""Per-stage error of the HIP LGM branch vs the oracle composition (GPU only; debugging aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videomv_amd.lgm import LgmRefiner, LgmOptions, lgm_param_shapes
from videomv_amd.registry import AUTO_ENCODER
from videomv_amd import ops
import videomv_amd
from oracle.lgm_ref import LgmCfg, forward_gaussians
from oracle.vae_ref import vae_decode, vae_encode_moments
from oracle.gs_ref import render_views
from oracle.weights import random_state_dict, vae_decoder_param_shapes, vae_encoder_param_shapes
from tests.test_gs_gpu import _cams

def rel(a, b): return float((a.float().cpu() - b.float().cpu()).norm() / b.float().cpu().norm())
c = dict(down_channels=(32, 64), down_attention=(False, True), mid_attention=True, up_channels=(64, 32),
         up_attention=(True, False), num_heads=2, input_size=64, splat_size=64, output_size=128)
opt = LgmOptions(**c); ocfg = LgmCfg(**c)
lsd = random_state_dict(lgm_param_shapes(opt), 808)
vsd = dict(random_state_dict(vae_decoder_param_shapes(ch=32), 77)); vsd.update(random_state_dict(vae_encoder_param_shapes(ch=32), 78))
dd = dict(double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)
vae = AUTO_ENCODER.build(dict(type="AutoencoderKL", ddconfig=dd, embed_dim=4)); vae.load_state_dict(vsd, strict=False)
g = torch.Generator().manual_seed(3)
z4 = torch.randn(4, 4, 8, 8, generator=g) * 3
rays = torch.randn(4, 6, 64, 64, generator=g)
cam_view, cam_vp = _cams(4, dist=2.2)
dev = torch.device("cuda", 0)
ref = LgmRefiner(opt, lsd, dev)
kw = dict(ch_mult=(1, 2, 4, 4), num_res_blocks=2)
dec_h = vae.decode(z4.to(dev)); dec_o = vae_decode(vsd, z4, **kw)
print("decode", rel(dec_h, dec_o), "range", float(dec_o.min()), float(dec_o.max()))
x = (dec_o * 0.5 + 0.5).clamp(0, 1)
mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1); std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
inp_o = torch.cat([(x - mean) / std, rays], dim=1)
inp_h = torch.zeros(4, 9, 64, 64, device=dev)
ops.lgm_pack_input(dec_o.to(dev).contiguous(), rays.to(dev).contiguous(), inp_h)
print("pack", rel(inp_h, inp_o))
ga_h = ref.engine.forward_gaussians(inp_o.to(dev)).clone(); ga_o = forward_gaussians(lsd, ocfg, inp_o.unsqueeze(0))[0]
print("gaussians", rel(ga_h, ga_o), "opacity mean", float(ga_o[:, 3].mean()), "scale mean", float(ga_o[:, 4:7].mean()))
bg = torch.full((3,), 0.5)
im_o, _ = render_views(ga_o, cam_view, cam_vp, 128, 39.6, bg)
out = ref.renderer.render(ga_o.to(dev).unsqueeze(0), cam_view.unsqueeze(0).to(dev), cam_vp.unsqueeze(0).to(dev), None, bg_color=bg.to(dev))
print("render(same gaussians)", rel(out["image"][0], im_o), "instances", ref.renderer.last_num_rendered)
out2 = ref.renderer.render(ga_h.unsqueeze(0), cam_view.unsqueeze(0).to(dev), cam_vp.unsqueeze(0).to(dev), None, bg_color=bg.to(dev))
print("render(hip gaussians)", rel(out2["image"][0], im_o), "image std", float(im_o.std()))
small_o = (im_o[:, :, ::2, ::2] - 0.5) / 0.5
mo_o = vae_encode_moments(vsd, small_o, **kw)
d = vae.encode(small_o.to(dev))
mo_h = d.moment_rows.view(4, 8, 8, -1)[..., :8].permute(0, 3, 1, 2)
print("moments(same images)", rel(mo_h, mo_o), "logvar range", float(mo_o[:, 4:].min()), float(mo_o[:, 4:].max()))

"""``UNetSD_I2VGen`` — drop-in for the reference image-to-multi-view UNet (tools/modules/unet/unet_i2vgen.py:27-404,
registry name ``MODEL: UNetSD_I2VGen``; BASELINE configs[3]) on the HIP hot path.

The trunk (encoder / middle / decoder / head) is the T2V trunk with an ``in_dim + concat_dim`` channel input conv and
145 context tokens; this file adds the I2VGen-XL *front-end*, which depends only on the conditioning image and is
therefore run ONCE per sample, not on each of the 100 forwards as the reference does:
  * ``local_image_concat`` (3 convs) + ``local_temporal_encoder`` (TransformerV2 over the frames of every pixel) ->
    ``concat`` written twice as in the reference (:345-346), stored in channels 4..7 of the input rows;
  * ``local_image_embedding`` (conv, adaptive-avg-pool 32x32, two stride-2 convs) -> 64 context tokens;
  * ``context_embedding`` (CLIP image feature -> ``num_tokens`` tokens); ``fps_embedding`` added to the time embedding.
Everything runs on libvmv_hip.so (implicit-GEMM convs, ``vmv_i2v_temporal_adapter``, ``vmv_adaptive_avgpool_rows``).
The LGM refinement branch (``autoencoder=`` + ``gs_data``, unet_i2vgen.py:438-468) is the shared ``LgmMixin`` of ``unet_t2v.py`` with the
v-prediction x0 (``lgm_vpred``); only its training-time form (``x0 is not None``) raises.
"""
import math
from typing import Dict

import os

import torch
import torch.nn as nn

from .registry import MODEL
from . import _lib as L
from . import ops
from . import packing as P
from .unet_engine import UNetEngine, param_shapes
from .unet_t2v import LgmMixin, _Holder, _ZERO_INIT_SUFFIXES, CondCache, same_for_both_branches


def i2v_extra_shapes(arch: dict, concat_dim: int, num_tokens: int, y_dim: int) -> Dict[str, tuple]:
    dim, ctx = arch["dim"], arch["context_dim"]
    E, cd = dim * 4, concat_dim
    s = [("context_embedding.0.weight", (E, y_dim)), ("context_embedding.0.bias", (E,)),
         ("context_embedding.2.weight", (ctx * num_tokens, E)), ("context_embedding.2.bias", (ctx * num_tokens,)),
         ("fps_embedding.0.weight", (E, dim)), ("fps_embedding.0.bias", (E,)),
         ("fps_embedding.2.weight", (E, E)), ("fps_embedding.2.bias", (E,)),
         ("local_image_concat.0.weight", (cd * 4, 4, 3, 3)), ("local_image_concat.0.bias", (cd * 4,)),
         ("local_image_concat.2.weight", (cd * 4, cd * 4, 3, 3)), ("local_image_concat.2.bias", (cd * 4,)),
         ("local_image_concat.4.weight", (cd, cd * 4, 3, 3)), ("local_image_concat.4.bias", (cd,)),
         ("local_temporal_encoder.layers.0.0.norm.weight", (cd,)), ("local_temporal_encoder.layers.0.0.norm.bias", (cd,)),
         ("local_temporal_encoder.layers.0.0.fn.to_qkv.weight", (2 * cd * 3, cd)),
         ("local_temporal_encoder.layers.0.0.fn.to_out.0.weight", (cd, 2 * cd)),
         ("local_temporal_encoder.layers.0.0.fn.to_out.0.bias", (cd,)),
         ("local_temporal_encoder.layers.0.1.net.0.0.weight", (cd * 4, cd)),
         ("local_temporal_encoder.layers.0.1.net.0.0.bias", (cd * 4,)),
         ("local_temporal_encoder.layers.0.1.net.2.weight", (cd, cd * 4)),
         ("local_temporal_encoder.layers.0.1.net.2.bias", (cd,)),
         ("local_image_embedding.0.weight", (cd * 8, 4, 3, 3)), ("local_image_embedding.0.bias", (cd * 8,)),
         ("local_image_embedding.3.weight", (cd * 16, cd * 8, 3, 3)), ("local_image_embedding.3.bias", (cd * 16,)),
         ("local_image_embedding.5.weight", (1024, cd * 16, 3, 3)), ("local_image_embedding.5.bias", (1024,))]
    return dict(s)


class I2VFrontEnd:
    """Once-per-sample image conditioning on the HIP kernels; writes into a ``UNetEngine``'s static buffers."""

    def __init__(self, sd: Dict[str, torch.Tensor], dim: int, num_tokens: int, device):
        self.dev, self.dim, self.E, self.num_tokens = device, dim, dim * 4, num_tokens
        w = {}
        for k in ("local_image_concat.0", "local_image_concat.2", "local_image_concat.4", "local_image_embedding.0",
                  "local_image_embedding.3", "local_image_embedding.5"):
            w[k + ".weight"] = P.pack_conv3x3(sd[k + ".weight"], device)
            w[k + ".bias"] = P.pack_bias(sd[k + ".bias"], device)
        for k in ("context_embedding.0", "context_embedding.2", "fps_embedding.0", "fps_embedding.2"):
            w[k + ".weight"] = P.pack_linear(sd[k + ".weight"], device)
            w[k + ".bias"] = P.pack_bias(sd[k + ".bias"], device)
        p = "local_temporal_encoder.layers.0"
        blob = [sd[f"{p}.0.norm.weight"], sd[f"{p}.0.norm.bias"], sd[f"{p}.0.fn.to_qkv.weight"],
                sd[f"{p}.0.fn.to_out.0.weight"], sd[f"{p}.0.fn.to_out.0.bias"], sd[f"{p}.1.net.0.0.weight"],
                sd[f"{p}.1.net.0.0.bias"], sd[f"{p}.1.net.2.weight"], sd[f"{p}.1.net.2.bias"]]
        w["adapter"] = torch.cat([t.detach().float().reshape(-1) for t in blob]).to(device).contiguous()
        assert w["adapter"].numel() == 288, "local_temporal_encoder must be TransformerV2(heads=2, dim=4, mlp 16)"
        self.w = w
        self._keep = []

    def _conv(self, S, x, cin, key, n, ih, iw, stride=1, silu=False):
        oh, ow = (ih + stride - 1) // stride, (iw + stride - 1) // stride
        W = self.w[key + ".weight"]
        y = torch.empty(n * oh * ow, W.shape[0], dtype=L.elem(), device=self.dev)
        S.gemm(ops.gemm_params(n * oh * ow, W.shape[0], ops.conv3x3_segs([(x, cin, cin)]), W, y, W.shape[0],
                               bias=self.w[key + ".bias"], act=L.ACT_SILU if silu else L.ACT_NONE,
                               geom=ops.Geom(OH=oh, OW=ow, IH=ih, IW=iw, stride=stride)), key)
        self._keep += [x, y]
        return y, oh, ow

    def _mlp(self, S, x_rows, key, out_fp32):
        n, k = x_rows.shape
        hid = torch.empty(n, self.E, dtype=L.elem(), device=self.dev)
        W0, W2 = self.w[key + ".0.weight"], self.w[key + ".2.weight"]
        S.gemm(ops.gemm_params(n, self.E, ops.linear_segs([(x_rows, k, k)]), W0, hid, self.E, bias=self.w[key + ".0.bias"],
                               act=L.ACT_SILU), key + ".0")
        out = torch.empty(n, W2.shape[0], dtype=torch.float32 if out_fp32 else L.elem(), device=self.dev)
        S.gemm(ops.gemm_params(n, W2.shape[0], ops.linear_segs([(hid, self.E, self.E)]), W2, out, W2.shape[0],
                               bias=self.w[key + ".2.bias"], out_fp32=out_fp32), key + ".2")
        self._keep += [x_rows, hid, out]
        return out

    @torch.no_grad()
    def run(self, eng: UNetEngine, local_images, ys, images, fps):
        """local_images [B,4,h,w] fp32, ys [B,Ly,ctx], images [B,1,y_dim] (one row per branch), fps [n_t].
        Fills eng.x_rows[:, 4:8], eng.ctx_rows and eng.extra_emb."""
        dev, B, F, h, w = self.dev, eng.B, eng.F, eng.H, eng.W
        S = ops.Stream(record=False)
        self._keep = []
        Ly = ys.shape[1]
        assert eng.L == Ly + 64 + self.num_tokens, "context length must be Ly + 64 local + num_tokens image tokens"
        ctx = eng.ctx_rows.view(B, eng.L, -1)
        ctx[:, :Ly].copy_(ys.to(L.elem()))
        img_tok = self._mlp(S, images.reshape(B, -1).to(L.elem()).contiguous(), "context_embedding", out_fp32=False)
        ctx[:, Ly + 64:].copy_(img_tok.view(B, self.num_tokens, -1))
        ld = eng.cin_pad
        for b in range(B):
            same = next((q for q in ((0, b - 1) if b > 0 else ()) if torch.equal(local_images[b], local_images[q])), None)
            if same is not None:                                             # CFG pair (block 0, or the previous block of a pair-major batch): same conditioning image
                eng.x_rows.view(B, -1, ld)[b, :, 4:8].copy_(eng.x_rows.view(B, -1, ld)[same, :, 4:8])
                ctx[b, Ly:Ly + 64].copy_(ctx[same, Ly:Ly + 64])
                continue
            li = local_images[b:b + 1].float()
            seq = torch.empty(1, 4, F, h, w, dtype=torch.float32, device=dev)                 # :333-335
            seq[:, :, 0] = li
            for i in range(1, F):
                seq[:, :, i] = i / (F - 1)
            rows_a = torch.zeros(F * h * w, 8, dtype=L.elem(), device=dev)
            ops.latent_to_rows_keep(seq, rows_a, 8, 1)
            a1, _, _ = self._conv(S, rows_a, 8, "local_image_concat.0", F, h, w, silu=True)
            a2, _, _ = self._conv(S, a1, a1.shape[1], "local_image_concat.2", F, h, w, silu=True)
            a3, _, _ = self._conv(S, a2, a2.shape[1], "local_image_concat.4", F, h, w)
            xr = eng.x_rows.view(B, -1, ld)[b]
            ops.i2v_temporal_adapter(a3, a3.shape[1], xr.data_ptr() + 8, ld, self.w["adapter"], F, h * w, 1, 2.0)
            rows_l = torch.zeros(h * w, 8, dtype=L.elem(), device=dev)
            ops.latent_to_rows_keep(li.reshape(1, 4, 1, h, w).contiguous(), rows_l, 8, 1)
            l1, _, _ = self._conv(S, rows_l, 8, "local_image_embedding.0", 1, h, w, silu=True)
            l2 = torch.empty(32 * 32, l1.shape[1], dtype=L.elem(), device=dev)
            ops.adaptive_avgpool_rows(l1, l1.shape[1], l2, l2.shape[1], 1, l1.shape[1], h, w, 32, 32)
            l3, oh, ow = self._conv(S, l2, l2.shape[1], "local_image_embedding.3", 1, 32, 32, stride=2, silu=True)
            l4, oh, ow = self._conv(S, l3, l3.shape[1], "local_image_embedding.5", 1, oh, ow, stride=2)
            ctx[b, Ly:Ly + 64].copy_(l4.view(64, -1))
            self._keep += [seq, rows_a, rows_l, l2]
        t_f = fps.to(dev).float().reshape(-1)
        sin = torch.empty(t_f.numel(), self.dim, dtype=L.elem(), device=dev)
        ops.sinusoidal(t_f, sin, t_f.numel(), self.dim)
        fe = self._mlp(S, sin, "fps_embedding", out_fp32=True)
        eng.extra_emb = fe if fe.shape[0] == eng.n_t else fe[:1].expand(eng.n_t, -1).contiguous()
        eng.context_updated()           # K / V of the (text + local-image + image) context tokens: once per sample
        if dev.type == "cuda":
            torch.cuda.synchronize()          # temporaries above are released when this returns
        self._keep = []


@MODEL.register_class()
class UNetSD_I2VGen(nn.Module, LgmMixin):
    lgm_bg_color = 0.7            # unet_i2vgen.py:458
    lgm_vpred = True              # unet_i2vgen.py:441-442

    def __init__(self, config=None, in_dim=4, dim=512, y_dim=512, context_dim=512, hist_dim=156, concat_dim=8,
                 dim_condition=4, out_dim=6, num_tokens=4, dim_mult=[1, 2, 3, 4], num_heads=None, head_dim=64,
                 num_res_blocks=3, attn_scales=[1 / 2, 1 / 4, 1 / 8], use_scale_shift_norm=True, dropout=0.1,
                 temporal_attn_times=1, temporal_attention=True, use_checkpoint=False, use_image_dataset=False,
                 use_sim_mask=False, training=True, inpainting=True, camera_dim=16, use_fps_condition=False,
                 use_camera_condition=False, use_lgm_refine=False, p_all_zero=0.1, p_all_keep=0.1, zero_y=None,
                 adapter_transformer_layers=1, lgm_opt=None, **kwargs):
        super().__init__()
        concat_dim = in_dim                           # the reference overrides the argument (unet_i2vgen.py:93)
        if concat_dim != 4 or adapter_transformer_layers != 1:
            raise NotImplementedError("I2VGen front-end is built for concat_dim = in_dim = 4, one adapter layer")
        num_heads = num_heads if num_heads else dim // 32
        self.zero_y, self.in_dim, self.dim, self.y_dim, self.context_dim = zero_y, in_dim, dim, y_dim, context_dim
        self.out_dim, self.num_tokens, self.concat_dim = out_dim, num_tokens, concat_dim
        self.use_camera_condition, self.use_lgm_refine, self.inpainting = use_camera_condition, use_lgm_refine, inpainting
        self.arch = dict(in_dim=in_dim + concat_dim, dim=dim, context_dim=context_dim, out_dim=out_dim,
                         dim_mult=list(dim_mult), num_heads=num_heads, head_dim=head_dim, num_res_blocks=num_res_blocks,
                         attn_scales=list(attn_scales), camera_dim=camera_dim, use_camera_condition=use_camera_condition,
                         use_fps_condition=False)
        shapes = dict(param_shapes(self.arch))
        shapes.update(i2v_extra_shapes(self.arch, concat_dim, num_tokens, y_dim))
        for key, shape in shapes.items():
            if key.endswith(_ZERO_INIT_SUFFIXES) or key == "out.2.weight" or key.startswith(("camera_embedding.2.", "fps_embedding.2.")):
                v = torch.zeros(shape)
            elif len(shape) == 1:
                v = torch.zeros(shape) if key.endswith(".bias") else torch.ones(shape)
            else:
                fan = 1
                for d in shape[1:]:
                    fan *= d
                v = torch.empty(shape).normal_(0.0, 1.0 / math.sqrt(fan))
            head, _, rest = key.partition(".")
            child = self._modules.get(head)
            if child is None:
                child = _Holder()
                self.add_module(head, child)
            child.add(rest, nn.Parameter(v, requires_grad=False))
        self._init_lgm(use_lgm_refine, lgm_opt)
        self._engines, self._front = {}, {}
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._invalidate())

    def _invalidate(self):
        self._lgm = None
        self._engines.clear()
        self._front.clear()

    def _get(self, B, F, H, W, L, device, n_t, share_prefix=False):
        share_prefix = bool(share_prefix) and os.environ.get("VMV_SHARE_PREFIX", "1") != "0"
        key = (B, F, H, W, L, str(device), n_t, share_prefix)
        if key not in self._engines:
            sd = {k: v.detach() for k, v in self.state_dict().items()}
            self._engines[key] = UNetEngine(self.arch, sd, B, F, H, W, L, device, n_t=n_t, share_prefix=share_prefix)
            if str(device) not in self._front:
                self._front[str(device)] = I2VFrontEnd(sd, self.dim, self.num_tokens, device)
        return self._engines[key], self._front[str(device)]

    @staticmethod
    def _first_frame(local_image):
        if local_image.ndim == 5:
            return local_image[:, :, 0]
        return local_image

    @torch.no_grad()
    def forward(self, x, t, x0=None, gs_data=None, sqrt_alphas_cumprod=None, sqrt_one_minus_alphas_cumprod=None,
                sqrt_recip_alphas_cumprod=None, sqrt_recipm1_alphas_cumprod=None, autoencoder=None, y=None, image=None,
                local_image=None, camera_data=None, masked=None, fps=None, video_mask=None, focus_present_mask=None,
                prob_focus_present=0., mask_last_frame_num=0, **kwargs):
        assert self.inpainting or masked is None, 'inpainting is not supported'
        if self.use_lgm_refine and x0 is not None:
            raise NotImplementedError("training-time LGM branch (unet_i2vgen.py:406-436)")
        if autoencoder is not None and not self.use_lgm_refine:
            raise ValueError("autoencoder=... needs a model built with use_lgm_refine=True")
        if local_image is None or image is None or y is None or fps is None:
            raise ValueError("UNetSD_I2VGen needs y, image, local_image and fps")
        b, c, f, h, w = x.shape
        dev = x.device
        L_ctx = y.shape[1] + 64 + self.num_tokens
        eng, front = self._get(b, f, h, w, L_ctx, dev, n_t=b)
        front.run(eng, self._first_frame(local_image).to(dev), y.to(dev).float(), image.to(dev).float(), fps)
        eng.set_camera(camera_data.to(dev) if (camera_data is not None and self.use_camera_condition) else None)
        eng.forward_rows(x.float(), t.to(dev))
        if autoencoder is None:
            return eng.eps_ncfhw()
        return self._lgm_branch(eng, x, t, autoencoder, gs_data, dict(
            sqrt_alphas_cumprod=sqrt_alphas_cumprod, sqrt_one_minus_alphas_cumprod=sqrt_one_minus_alphas_cumprod,
            sqrt_recip_alphas_cumprod=sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod=sqrt_recipm1_alphas_cumprod))

    @torch.no_grad()
    def forward_cfg_rows(self, xt, t, cond_kwargs, uncond_kwargs):
        """cond / uncond branches in one pass; image conditioning is evaluated once per sample (cached on the tensors'
        identity).  kwargs as in inference_i2vgen_entrance.py:267-269: y, image, local_image, fps, camera_data."""
        b, c, f, h, w = xt.shape
        dev = xt.device
        kc, ku = cond_kwargs, uncond_kwargs
        if ku.get("image") is None:
            raise NotImplementedError("uncond branch without image tokens (use_zero_infer=False) has a shorter context")
        if b != 1:
            return self._forward_cfg_rows_batched(xt, t, kc, ku)
        L_ctx = kc["y"].shape[1] + 64 + self.num_tokens
        # (local_image / fps / camera are checked to be the same for both branches below: the CFG pair shares its prefix)
        eng, front = self._get(2, f, h, w, L_ctx, dev, n_t=1, share_prefix=True)
        cache = eng.__dict__.setdefault("_cond", CondCache())
        if not cache.hit(kc["y"], ku["y"], kc["image"], ku["image"], kc["local_image"], kc["fps"], kc.get("camera_data"),
                         ku.get("local_image"), ku.get("fps"), ku.get("camera_data")):
            # (once per sample, not per step: each comparison is a device-to-host sync — ADVICE r2)
            same_for_both_branches("local_image", kc.get("local_image"), ku.get("local_image"))
            same_for_both_branches("fps", kc.get("fps"), ku.get("fps"))
            same_for_both_branches("camera_data", kc.get("camera_data"), ku.get("camera_data"))
            li = self._first_frame(kc["local_image"]).to(dev)
            front.run(eng, torch.cat([li, li], dim=0), torch.cat([kc["y"], ku["y"]], dim=0).to(dev).float(),
                      torch.cat([kc["image"], ku["image"]], dim=0).to(dev).float(), kc["fps"][:1])
            cam = kc.get("camera_data")
            eng.set_camera(cam.to(dev) if (cam is not None and self.use_camera_condition) else None)
        return eng, eng.forward_rows(xt.float(), t.to(dev))

    cfg_batch_ok = True       # DiffusionDDIM.ddim_sample_loop: noise [b > 1, ...] may take the fused path

    @torch.no_grad()
    def _forward_cfg_rows_batched(self, xt, t, kc, ku):
        """b > 1 input images of one denoising step in ONE plan of B = 2 b row blocks, pair-major [c_0 | u_0 | c_1 | u_1 ...] (round 6; as
        unet_t2v._forward_cfg_rows_batched — the reference's sampler API admits the batch, inference_i2vgen_entrance.py:267-270 feeds it
        one image at a time).  Per-sample: y, image, local_image; shared ([1, ...]) or per-sample: the uncond branch's y / image,
        camera_data; fps is one value.  With one camera set the CFG prefix is shared per sample (the front-end's image channels of sample s are
        moved to row block s, which the prefix reads)."""
        b, c, f, h, w = xt.shape
        dev = xt.device
        L_ctx = kc["y"].shape[1] + 64 + self.num_tokens
        cam_in = kc.get("camera_data")
        one_cam = (not self.use_camera_condition) or cam_in is None or cam_in.numel() == f * cam_in.shape[-1]
        # (local_image / fps / camera are checked to be the same for both branches below; with ONE camera set the CFG prefix is recorded
        #  on one row block per sample and replicated pairwise, as in the single-image pass)
        eng, front = self._get(2 * b, f, h, w, L_ctx, dev, n_t=1, share_prefix=one_cam)
        cache = eng.__dict__.setdefault("_cond", CondCache())
        if not cache.hit(kc["y"], ku["y"], kc["image"], ku["image"], kc["local_image"], kc["fps"], kc.get("camera_data"),
                         ku.get("local_image"), ku.get("fps"), ku.get("camera_data")):
            same_for_both_branches("local_image", kc.get("local_image"), ku.get("local_image"))
            same_for_both_branches("fps", kc.get("fps"), ku.get("fps"))
            same_for_both_branches("camera_data", kc.get("camera_data"), ku.get("camera_data"))

            def per_sample(v, what):
                v = v.to(dev).float()
                if v.shape[0] not in (1, b):
                    raise ValueError(f"{what} has batch {v.shape[0]}, expected 1 or {b}")
                return v.expand(b, *v.shape[1:])
            pair = lambda a_, b_: torch.stack([a_, b_], dim=1).reshape(2 * b, *a_.shape[1:]).contiguous()       # pair-major row blocks
            li = per_sample(self._first_frame(kc["local_image"]), "local_image")
            front.run(eng, pair(li, li), pair(per_sample(kc["y"], "cond y"), per_sample(ku["y"], "uncond y")),
                      pair(per_sample(kc["image"], "cond image"), per_sample(ku["image"], "uncond image")), kc["fps"][:1])
            if eng.share_prefix:
                # the shared prefix reads ONE row block per sample, [s_0 | s_1 ...]: move sample s's image channels from its pair's first
                # block (2 s) to block s (ascending: a source 2 s is never an earlier destination)
                xr = eng.x_rows.view(2 * b, -1, eng.cin_pad)
                for s_ in range(1, b):
                    xr[s_, :, 4:8].copy_(xr[2 * s_, :, 4:8])
            cam = kc.get("camera_data")
            if cam is not None and self.use_camera_condition:
                cam = cam.to(dev).float().reshape(-1, f, cam.shape[-1])
                cam = cam if cam.shape[0] == 1 else pair(per_sample(cam, "camera_data"), per_sample(cam, "camera_data"))
            else:
                cam = None
            eng.set_camera(cam)
        eng.prepare_rows(xt.float(), t.to(dev), pair_major=True)
        eng.run_plan()
        return eng, eng.eps_rows

    def begin_sample(self):
        """Drop the per-sample conditioning caches (called by the sampler at the start of every ddim_sample_loop)."""
        for eng in self._engines.values():
            if "_cond" in eng.__dict__:
                eng._cond.clear()

"""``UNetSD_T2VBase`` — drop-in for the reference module of the same name (tools/modules/unet/unet_t2v.py:55-433)
whose forward runs on the hand-written gfx950 kernels of ``libvmv_hip.so``.

What stays identical to the reference:
  * registry name ``MODEL: UNetSD_T2VBase`` and the constructor keywords (python defaults + YAML keys);
  * the ``state_dict`` key names/shapes (1484 keys at full size, typos included — SURVEY F13), so reference
    checkpoints load with ``load_state_dict(sd, strict=False)``;
  * the ``forward`` signature and the output layout/dtype (``[b, out_dim, f, h, w]``, fp32).
What differs by design: there is no PyTorch math in ``forward``.  Parameters are repacked to 16-bit GEMM
operands on first use and the whole forward is a recorded plan of HIP launches (``unet_engine.UNetEngine``).
If the HIP library is missing, construction of the engine raises — there is no CPU fallback.

The LGM refinement branch (``autoencoder is not None``, unet_t2v.py:404-433; ``use_lgm_refine=True`` registers the
``lgm_big.*`` parameters, :125-129) runs on ``lgm.LgmRefiner``; only its training-time variant (``x0 is not None``,
:370-400, which needs ground-truth renders and the LGM losses) raises ``NotImplementedError``.
"""
import math
import os
from typing import Dict, Optional

import torch
import torch.nn as nn

from .registry import MODEL
from .unet_engine import UNetEngine, param_shapes, block_plan

_ZERO_INIT_SUFFIXES = ("proj_out.weight", "proj_out.bias", "out_layers.3.weight", "out_layers.3.bias",
                       "temopral_conv.conv4.3.weight", "temopral_conv.conv4.3.bias")


class _Holder(nn.Module):
    """Parameter container that reproduces the reference's dotted state-dict names."""

    def add(self, dotted: str, param: nn.Parameter):
        head, _, rest = dotted.partition(".")
        if not rest:
            self.register_parameter(head, param)
            return
        child = self._modules.get(head)
        if child is None:
            child = _Holder()
            self.add_module(head, child)
        child.add(rest, param)


class LgmMixin:
    """``self.lgm_big = LGM(config_defaults['big'])`` of the video UNets (unet_t2v.py:125-129, unet_i2vgen.py:108-112): the
    376 ``lgm_big.*`` parameters (415 M) and the inference branch that returns latent_z instead of eps."""
    lgm_bg_color = 0.5            # LGM.infer(bg_color_factor): 0.5 for T2V (models.py:115), 0.7 for I2VGen (unet_i2vgen.py:458)
    lgm_vpred = False             # x0 from eps (unet_t2v.py:405) or from v (unet_i2vgen.py:441-442)

    def _init_lgm(self, use_lgm_refine, lgm_opt):
        self.lgm_opt, self._lgm = None, None
        if not use_lgm_refine:
            return
        from .lgm import LgmOptions, lgm_param_shapes
        self.lgm_opt = LgmOptions(**lgm_opt) if isinstance(lgm_opt, dict) else (lgm_opt or LgmOptions())
        for key, shape in lgm_param_shapes(self.lgm_opt).items():
            if len(shape) == 1:
                v = torch.zeros(shape) if key.endswith(".bias") else torch.ones(shape)
            else:
                fan_in = 1
                for d in shape[1:]:
                    fan_in *= d
                v = torch.empty(shape).normal_(0.0, 1.0 / math.sqrt(fan_in))
            head, _, rest = ("lgm_big." + key).partition(".")
            child = self._modules.get(head)
            if child is None:
                child = _Holder()
                self.add_module(head, child)
            child.add(rest, nn.Parameter(v, requires_grad=False))

    def lgm_refiner(self, device):
        if not self.use_lgm_refine:
            raise ValueError("model was built with use_lgm_refine=False")
        if self._lgm is None:
            from .lgm import LgmRefiner
            sd = {k[len("lgm_big."):]: v.detach() for k, v in self.state_dict().items() if k.startswith("lgm_big.")}
            self._lgm = LgmRefiner(self.lgm_opt, sd, device, bg_color=self.lgm_bg_color)
        return self._lgm

    def _lgm_branch(self, eng, x, t, autoencoder, gs_data, tables):
        """eps / v rows of ONE sample in ``eng`` -> latent_z (the value forward() returns when ``autoencoder`` is given)."""
        if x.shape[0] != 1:
            raise ValueError("the LGM branch handles one sample per call (the reference's gs_data has batch 1)")
        step = int(t.reshape(-1)[0])
        a, b = ("sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod") if self.lgm_vpred else \
               ("sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod")
        if tables.get(a) is None or tables.get(b) is None:
            raise ValueError(f"the LGM branch needs the schedule tables {a} / {b}")
        ca, cb = float(tables[a][step].to(torch.float32)), float(tables[b][step].to(torch.float32))
        return self.lgm_refiner(x.device).latent_z(eng.eps_rows, eng.out_pad, 0, x.float().contiguous(), ca, cb, autoencoder,
                                                   dict(gs_data))


def gather_frames(comm, x_local: torch.Tensor) -> torch.Tensor:
    """[b, c, F/R, h, w] per rank -> [b, c, F, h, w] on every rank (frame-parallel sampling)."""
    x_local = x_local.contiguous()
    out = torch.empty((comm.world,) + tuple(x_local.shape), dtype=x_local.dtype, device=x_local.device)
    comm.all_gather(out.view(comm.world, -1), x_local.view(-1))
    return torch.cat(list(out.unbind(0)), dim=2)


class _PipeHandle:
    """What the fused sampler needs from "the engine" of a forward_cfg_rows call (the eps row pitch)."""

    def __init__(self, out_pad):
        self.out_pad = out_pad


class CondCache:
    """Step-invariant conditioning of the fused CFG pass (context rows, camera MLP, I2V image front-end) is evaluated once
    per sample.  The cache key is the IDENTITY and in-place VERSION of the conditioning tensors, and the cache holds
    strong references to them, so neither a freed-and-reallocated buffer (same address, new contents) nor an in-place
    refill of the same tensor can be mistaken for the cached sample.  ``DiffusionDDIM.ddim_sample_loop`` additionally
    drops it at the start of every sample (``begin_sample``)."""

    def __init__(self):
        self.key, self.refs = None, None

    def hit(self, *tensors) -> bool:
        key = tuple(None if t is None else (id(t), t._version) for t in tensors)
        if key == self.key:
            return True
        self.key, self.refs = key, tensors
        return False

    def clear(self):
        self.key, self.refs = None, None


def same_for_both_branches(name, a, b):
    """The fused pass evaluates shared conditioning once: the two model_kwargs dicts must agree on it (the reference
    evaluates each branch with its own kwargs, diffusion_ddim.py:149-156)."""
    if a is b:
        return
    if (a is None) != (b is None) or a.shape != b.shape or not torch.equal(a.to(b.device), b):
        raise NotImplementedError(f"cond / uncond model_kwargs differ in `{name}`: not supported by the fused CFG pass "
                                  f"(call model.forward per branch through DiffusionDDIM.ddim_sample instead)")


@MODEL.register_class()
class UNetSD_T2VBase(nn.Module, LgmMixin):
    def __init__(self, config=None, in_dim=4, dim=512, y_dim=512, context_dim=512, hist_dim=156, dim_condition=4,
                 out_dim=6, num_tokens=4, dim_mult=[1, 2, 3, 4], num_heads=None, head_dim=64, camera_dim=16,
                 num_res_blocks=3, attn_scales=[1 / 2, 1 / 4, 1 / 8], use_scale_shift_norm=True, dropout=0.1,
                 temporal_attn_times=1, temporal_attention=True, use_checkpoint=False, use_image_dataset=False,
                 use_sim_mask=False, training=True, inpainting=True, use_fps_condition=False,
                 use_camera_condition=False, use_lgm_refine=False, p_all_zero=0.1, p_all_keep=0.1, zero_y=None,
                 adapter_transformer_layers=1, lgm_opt=None, **kwargs):
        super().__init__()
        if not temporal_attention:
            raise NotImplementedError("temporal_attention=False is not a VideoMV configuration")
        num_heads = num_heads if num_heads else dim // 32
        self.zero_y = zero_y
        self.in_dim, self.dim, self.y_dim, self.context_dim = in_dim, dim, y_dim, context_dim
        self.out_dim, self.dim_mult, self.num_heads, self.head_dim = out_dim, list(dim_mult), num_heads, head_dim
        self.num_res_blocks, self.attn_scales = num_res_blocks, list(attn_scales)
        self.use_camera_condition, self.camera_dim = use_camera_condition, camera_dim
        self.use_fps_condition = use_fps_condition
        self.use_lgm_refine = use_lgm_refine
        self.inpainting = inpainting
        self.arch = dict(in_dim=in_dim, dim=dim, context_dim=context_dim, out_dim=out_dim, dim_mult=list(dim_mult),
                         num_heads=num_heads, head_dim=head_dim, num_res_blocks=num_res_blocks,
                         attn_scales=list(attn_scales), camera_dim=camera_dim,
                         use_camera_condition=use_camera_condition, use_fps_condition=use_fps_condition)
        for key, shape in param_shapes(self.arch).items():
            # the layers the reference zero-initialises: a fresh model is ~identity there too
            if key.endswith(_ZERO_INIT_SUFFIXES) or key == "out.2.weight" or key.startswith("camera_embedding.2."):
                v = torch.zeros(shape)
            elif len(shape) == 1:
                v = torch.zeros(shape) if key.endswith(".bias") else torch.ones(shape)
            else:
                fan_in = 1
                for d in shape[1:]:
                    fan_in *= d
                v = torch.empty(shape).normal_(0.0, 1.0 / math.sqrt(fan_in))
            self._add_param(key, nn.Parameter(v, requires_grad=False))
        self._init_lgm(use_lgm_refine, lgm_opt)
        self._engines: Dict[tuple, UNetEngine] = {}
        self._pipe = None             # the two branch engines of the pipelined frame-parallel mode
        self._packed_donor = None     # packed weights of an engine dropped by set_frame_parallel (reused by the next ones)
        self.frame_comm = None        # comm.FrameComm: frame-parallel execution over the ranks of one sample
        self._weights_version = 0
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._invalidate())

    def _add_param(self, dotted, param):
        head, _, rest = dotted.partition(".")
        child = self._modules.get(head)
        if child is None:
            child = _Holder()
            self.add_module(head, child)
        child.add(rest, param)

    def _invalidate(self):
        self._lgm = None
        self._engines.clear()
        self._pipe = None
        self._packed_donor = None
        self._weights_version += 1

    # ------------------------------------------------------------------ engine management
    def set_frame_parallel(self, comm):
        """Shard the frames of every sample over the ranks of ``comm`` (``comm.FrameComm``; None = off).  BASELINE
        configs[2]: 24 views, 3 per GPU.  ``forward`` keeps taking / returning whole samples; the fused sampler
        (``forward_cfg_rows``) works on this rank's frames.  No reference counterpart (its multi-GPU mode is replicas)."""
        self.frame_comm = comm
        donor = next(iter(self._engines.values()), None) or ((self._pipe or {}).get("engs") or [None])[0]
        if donor is not None:             # packed weights are shape- and sharding-independent: the next engines reuse them
            self._packed_donor = donor.packed
        self._engines.clear()
        self._pipe = None

    def engine_for(self, B, F, H, W, L, device, n_t=1, taps=None, share_prefix=False) -> UNetEngine:
        """F = frames of the whole sample.  share_prefix: the B = 2 branches are a CFG pair with identical x_t / t / camera /
        fps (UNetEngine.__init__)."""
        share_prefix = bool(share_prefix) and os.environ.get("VMV_SHARE_PREFIX", "1") != "0"
        key = (B, F, H, W, L, str(device), n_t, taps is not None, share_prefix)
        eng = self._engines.get(key)
        if eng is None:
            sd = {k: v.detach() for k, v in self.state_dict().items()}
            donor = next(iter(self._engines.values()), None)       # packed weights are shared by all engines of the model
            eng = UNetEngine(self.arch, sd, B, F, H, W, L, device, n_t=n_t, taps=taps, comm=self.frame_comm,
                             share_prefix=share_prefix, packed=donor.packed if donor is not None else self._packed_donor)
            if taps is None:
                self._engines[key] = eng
        return eng

    # ------------------------------------------------------------------ reference forward signature
    @torch.no_grad()
    def forward(self, x, t, x0=None, gs_data=None, sqrt_alphas_cumprod=None, sqrt_one_minus_alphas_cumprod=None,
                sqrt_recip_alphas_cumprod=None, sqrt_recipm1_alphas_cumprod=None, autoencoder=None, y=None, fps=None,
                masked=None, camera_data=None, video_mask=None, focus_present_mask=None, prob_focus_present=0.,
                mask_last_frame_num=0, **kwargs):
        assert self.inpainting or masked is None, 'inpainting is not supported'
        if self.use_lgm_refine and x0 is not None:
            raise NotImplementedError("training-time LGM branch (unet_t2v.py:370-400: ground-truth renders + LGM losses)")
        if autoencoder is not None and not self.use_lgm_refine:
            raise ValueError("autoencoder=... needs a model built with use_lgm_refine=True")
        b, c, f, h, w = x.shape
        dev = x.device
        if y is None:
            if self.zero_y is None:
                raise ValueError("y is None and no zero_y was given")
            y = self.zero_y.repeat(b, 1, 1)[:, :1, :]
        # the reference relies on DDP's scatter to move CPU kwargs: do it ourselves
        y = y.to(dev)
        if y.shape[0] == 1 and b > 1:        # (one text for the whole batch, e.g. the negative prompt: the reference wants it repeated by the caller)
            y = y.expand(b, -1, -1)
        if camera_data is not None:
            camera_data = camera_data.to(dev)
        eng = self.engine_for(b, f, h, w, y.shape[1], dev, n_t=b)
        eng.set_context(y.float())
        eng.set_camera(camera_data if self.use_camera_condition else None)
        eng.set_fps(fps if self.use_fps_condition else None)
        if self.frame_comm is None:
            eng.forward_rows(x.float(), t.to(dev))
            if autoencoder is None:
                return eng.eps_ncfhw()
            # LGM branch: eps -> x0 of 4 views -> LGM -> 24 renders -> latent_z (returned in place of eps)
            return self._lgm_branch(eng, x, t, autoencoder, gs_data, dict(
                sqrt_alphas_cumprod=sqrt_alphas_cumprod, sqrt_one_minus_alphas_cumprod=sqrt_one_minus_alphas_cumprod,
                sqrt_recip_alphas_cumprod=sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod=sqrt_recipm1_alphas_cumprod))
        comm = self.frame_comm                    # whole sample in, whole sample out: run this rank's frames, gather
        fl = f // comm.world
        eng.forward_rows(x[:, :, comm.rank * fl:(comm.rank + 1) * fl].float().contiguous(), t.to(dev))
        return gather_frames(comm, eng.eps_ncfhw())

    @torch.no_grad()
    def forward_cfg_rows(self, xt, t, cond_kwargs, uncond_kwargs):
        """(Frame-parallel: ``xt`` holds THIS rank's frames only; camera_data / y stay whole-sample.)
        Both classifier-free-guidance branches in ONE pass (B = 2 rows blocks sharing x_t, so weights stream once
        per step instead of twice — SURVEY App. C).  ``cond_kwargs`` / ``uncond_kwargs`` are the two ``model_kwargs``
        dicts of ``ddim_sample_loop`` (keys ``y``, ``camera_data``, ``fps`` — the latter ignored, as in the reference, when
        ``use_fps_condition`` is False).  Returns (engine, eps_rows fp32 [2*F*H*W, out_pad]); rows [0, F*H*W) are the
        conditional branch."""
        y_cond, y_uncond = cond_kwargs["y"], uncond_kwargs["y"]
        camera_data = cond_kwargs.get("camera_data", None)
        cam_u = uncond_kwargs.get("camera_data", None)
        b, c, f, h, w = xt.shape
        dev = xt.device
        if b != 1:
            if self.frame_comm is not None and hasattr(self.frame_comm, "exchange_branches"):
                raise ValueError("forward_cfg_rows over a CFG-parallel group handles one sample (the reference's noise is [1,4,F,h,w])")
            return self._forward_cfg_rows_batched(xt, t, cond_kwargs, uncond_kwargs)
        if self.frame_comm is not None:
            if hasattr(self.frame_comm, "exchange_branches"):          # comm.CfgFrameComm: one branch per rank group
                return self._forward_cfg_rows_cfgpar(xt, t, cond_kwargs, uncond_kwargs)
            if os.environ.get("VMV_FP_PIPELINE", "1") != "0":
                return self._forward_cfg_rows_pipelined(xt, t, cond_kwargs, uncond_kwargs)
            f = f * self.frame_comm.world
        cam_shared = self._cameras_agree(camera_data, cam_u)
        eng = self.engine_for(2, f, h, w, y_cond.shape[1], dev, n_t=1, share_prefix=cam_shared)
        cache = eng.__dict__.setdefault("_cond", CondCache())
        if not cache.hit(y_cond, y_uncond, camera_data, cam_u, cond_kwargs.get("fps")):      # context / camera are step-invariant: once per sample
            cam = None
            if self.use_camera_condition and (camera_data is not None or cam_u is not None):
                if camera_data is None or cam_u is None:
                    raise NotImplementedError("camera_data on only one CFG branch is not supported by the fused pass")
                cam = camera_data.to(dev)
                if cam_u is not camera_data and not (cam_u.shape == camera_data.shape and torch.equal(cam_u.to(dev), cam)):
                    cam = torch.cat([cam.reshape(1, -1, cam.shape[-1]), cam_u.to(dev).reshape(1, -1, cam.shape[-1])], dim=0)
            eng.set_context(torch.cat([y_cond.to(dev).float(), y_uncond.to(dev).float()], dim=0))
            eng.set_camera(cam)
            fps = cond_kwargs.get("fps") if self.use_fps_condition else None
            if self.use_fps_condition:      # (either branch alone having an fps is a disagreement too)
                same_for_both_branches("fps", fps, uncond_kwargs.get("fps"))
            eng.set_fps(fps)
        return eng, eng.forward_rows(xt.float(), t.to(dev))

    cfg_batch_ok = True       # DiffusionDDIM.ddim_sample_loop: noise [b > 1, ...] may take the fused path (_forward_cfg_rows_batched)

    @torch.no_grad()
    def _forward_cfg_rows_batched(self, xt, t, cond_kwargs, uncond_kwargs):
        """b > 1 samples (prompts) of one denoising step in ONE plan of B = 2 b row blocks, PAIR-major [c_0 | u_0 | c_1 | u_1 | ...]
        so that every sample's (cond, uncond) eps rows sit like the single-sample pass's and the fused CFG + DDIM update runs per
        sample on a row offset.  No reference counterpart in the entrance (it denoises one prompt at a time,
        inference_text2video_entrance.py:152-214) but the reference's sampler API admits it: noise [b, 4, F, h, w] with
        model_kwargs y [b, L, D] (diffusion_ddim.py:247-260, p_mean_variance :149-160).  Why: the small levels of one sample do
        not fill 256 CUs (a 24 x 32 x 32 step is 0.75 / 0.19 / 0.05 / 0.01 rounds of 256-row tiles at the four levels); two prompts in one plan
        measure 41.6 ms against 2 x 25.3 at that shape and 95.1 against 2 x 50.2 at 24 x 40 x 64 (profiles/r6_batch_fill.log).
        y [1, L, D] / camera_data [1, F, 16] are shared by all samples; with ONE camera set the CFG prefix (everything before the first
        cross-attention) is recorded on one row block per prompt and replicated pairwise, as in the 1-prompt pass."""
        y_c, y_u = cond_kwargs["y"], uncond_kwargs["y"]
        cam_c, cam_u = cond_kwargs.get("camera_data", None), uncond_kwargs.get("camera_data", None)
        b, c, f, h, w = xt.shape
        dev = xt.device
        if self.frame_comm is not None:       # frame-parallel (plain FrameComm, single plan): xt holds this rank's frames of all b samples;
            f = f * self.frame_comm.world     # a rank's GEMMs see b times the rows of its 1 / world share (comm engines never share a prefix)
        # one camera set [1, F, 16] for every row block (the entrance's orbit) => the CFG prefix is shared per prompt, as in the 1-prompt pass
        one_cam = (not self.use_camera_condition) or (cam_c is None and cam_u is None) or (
            cam_c is not None and cam_u is not None and cam_c.numel() == f * cam_c.shape[-1] and self._cameras_agree(cam_c, cam_u))
        eng = self.engine_for(2 * b, f, h, w, y_c.shape[1], dev, n_t=1, share_prefix=one_cam)
        cache = eng.__dict__.setdefault("_cond", CondCache())
        if not cache.hit(y_c, y_u, cam_c, cam_u, cond_kwargs.get("fps")):
            def per_sample(v, what):
                v = v.to(dev).float()
                v = v.reshape(-1, *v.shape[-2:])
                if v.shape[0] not in (1, b):
                    raise ValueError(f"{what} has batch {v.shape[0]}, expected 1 or {b}")
                return v.expand(b, *v.shape[1:])
            ctx = torch.stack([per_sample(y_c, "cond y"), per_sample(y_u, "uncond y")], dim=1)          # [b, 2, L, D]: pair-major
            eng.set_context(ctx.reshape(2 * b, *ctx.shape[2:]).contiguous())
            cam = None
            if self.use_camera_condition and (cam_c is not None or cam_u is not None):
                if cam_c is None or cam_u is None:
                    raise NotImplementedError("camera_data on only one CFG branch is not supported by the fused pass")
                cc, cu = cam_c.to(dev).float(), cam_u.to(dev).float()
                cc, cu = cc.reshape(-1, f, cc.shape[-1]), cu.reshape(-1, f, cu.shape[-1])
                if cc.shape[0] == 1 and cu.shape[0] == 1 and torch.equal(cc, cu):
                    cam = cc                                                                  # one camera set for every row block
                else:
                    cam = torch.stack([per_sample(cc, "cond camera_data"), per_sample(cu, "uncond camera_data")], dim=1).reshape(2 * b, f, -1)
            eng.set_camera(cam)
            fps = cond_kwargs.get("fps") if self.use_fps_condition else None
            if self.use_fps_condition:
                same_for_both_branches("fps", fps, uncond_kwargs.get("fps"))
            eng.set_fps(fps)
        eng.prepare_rows(xt.float(), t.to(dev), pair_major=True)
        eng.run_plan()
        return eng, eng.eps_rows

    def _cameras_agree(self, cam_c, cam_u) -> bool:
        """Do the two CFG branches carry the same camera_data (=> shared-prefix engine)?  The comparison is a device-to-host
        sync, so it is evaluated once per pair of tensors (identity + in-place version, strong references held), not on every
        denoising step."""
        if not self.use_camera_condition or cam_u is cam_c:
            return True
        if cam_u is None or cam_c is None:
            return False
        key = (id(cam_c), cam_c._version, id(cam_u), cam_u._version)
        memo = getattr(self, "_cam_agree", None)
        if memo is None or memo[0] != key:
            same = cam_u.shape == cam_c.shape and bool(torch.equal(cam_u.to(cam_c.device), cam_c))
            self._cam_agree = memo = (key, same, (cam_c, cam_u))
        return memo[1]

    def begin_sample(self):
        """Drop the per-sample conditioning caches (called by the sampler at the start of every ddim_sample_loop)."""
        for eng in self._engines.values():
            if "_cond" in eng.__dict__:
                eng._cond.clear()
        if getattr(self, "_pipe", None) is not None:
            self._pipe["cond"].clear()

    # ------------------------------------------------------------------ CFG-parallel x frame-parallel
    @torch.no_grad()
    def _forward_cfg_rows_cfgpar(self, xt, t, cond_kwargs, uncond_kwargs):
        """comm.CfgFrameComm: this rank runs ONE branch (cond for the first half of the ranks, uncond for the second) on its
        frames of that branch group, then swaps eps rows with the rank that ran the other branch on the same frames."""
        comm = self.frame_comm
        b, c, fl, h, w = xt.shape
        dev = xt.device
        F_all = fl * comm.world
        kw = (cond_kwargs, uncond_kwargs)[comm.branch]
        y, cam = kw["y"], kw.get("camera_data")
        key = ("cfgpar", F_all, h, w, y.shape[1], str(dev))
        pipe = getattr(self, "_pipe", None)
        if pipe is None or pipe["key"] != key or pipe["comm"] is not comm:
            sd = {k: v.detach() for k, v in self.state_dict().items()}
            T1 = fl * h * w
            out_pad = (self.out_dim + 3) // 4 * 4
            eps = torch.zeros(2 * T1, out_pad, dtype=torch.float32, device=dev)
            mine = torch.zeros(T1, out_pad, dtype=torch.float32, device=dev)
            eng = UNetEngine(self.arch, sd, 1, F_all, h, w, y.shape[1], dev, n_t=1, comm=comm.fp, eps_out=mine, packed=self._packed_donor)
            pipe = dict(key=key, comm=comm, engs=[eng], eps=eps, mine=mine, cond=CondCache(), out_pad=out_pad)
            self._pipe = pipe
        eng = pipe["engs"][0]
        if not pipe["cond"].hit(y, cam, kw.get("fps")):
            eng.set_context(y.to(dev).float())
            eng.set_camera(cam.to(dev) if (cam is not None and self.use_camera_condition) else None)
            eng.set_fps(kw.get("fps") if self.use_fps_condition else None)
        eng.forward_rows(xt.float(), t.to(dev))
        comm.exchange_branches(pipe["eps"].view(2, -1), pipe["mine"].view(-1))
        return _PipeHandle(pipe["out_pad"]), pipe["eps"]

    # ------------------------------------------------------------------ frame-parallel, branch-pipelined
    @torch.no_grad()
    def _forward_cfg_rows_pipelined(self, xt, t, cond_kwargs, uncond_kwargs):
        """Frame-parallel CFG pass as TWO one-branch plans on two streams, one communicator each: while one
        branch waits for a layout-switch all-to-all or a gathered GroupNorm sum, the other branch's kernels own the GPU, so
        the 183 latency-bound collectives of a step hide behind compute instead of serialising with it; with B = 1 every
        layout switch also needs only ONE of its two permute copies (UNetEngine._switch).  The two plans write the two
        halves of one eps buffer, which the fused CFG + DDIM kernel reads as before.  Weights are packed once and shared."""
        comm = self.frame_comm
        b, c, fl, h, w = xt.shape
        dev = xt.device
        F_all = fl * comm.world
        ys = (cond_kwargs["y"], uncond_kwargs["y"])
        cams = (cond_kwargs.get("camera_data"), uncond_kwargs.get("camera_data"))
        Lc = ys[0].shape[1]
        key = (F_all, h, w, Lc, str(dev))
        pipe = getattr(self, "_pipe", None)
        if pipe is None or pipe["key"] != key or pipe["comm"] is not comm:
            sd = {k: v.detach() for k, v in self.state_dict().items()}
            T1 = fl * h * w
            out_pad = (self.out_dim + 3) // 4 * 4            # rows of the packed head conv (packing._pad_rows)
            eps = torch.zeros(2 * T1, out_pad, dtype=torch.float32, device=dev)
            engs, comms = [], (comm, comm.twin())
            for br in range(2):
                engs.append(UNetEngine(self.arch, sd, 1, F_all, h, w, Lc, dev, n_t=1, comm=comms[br],
                                       packed=engs[0].packed if engs else self._packed_donor, eps_out=eps[br * T1:(br + 1) * T1]))
            streams = [torch.cuda.Stream(device=dev) for _ in range(2)] if dev.type == "cuda" else [None, None]
            pipe = dict(key=key, comm=comm, engs=engs, eps=eps, streams=streams, cond=CondCache(), out_pad=engs[0].out_pad)
            self._pipe = pipe
        engs, streams = pipe["engs"], pipe["streams"]
        if not pipe["cond"].hit(ys[0], ys[1], cams[0], cams[1], cond_kwargs.get("fps")):
            for br in range(2):
                engs[br].set_context(ys[br].to(dev).float())
                cam = cams[br]
                engs[br].set_camera(cam.to(dev) if (cam is not None and self.use_camera_condition) else None)
                engs[br].set_fps((cond_kwargs, uncond_kwargs)[br].get("fps") if self.use_fps_condition else None)
        x32, t_dev = xt.float(), t.to(dev)
        if streams[0] is None:                       # CPU (gloo tests): same schedule, no streams
            for br in range(2):
                engs[br].prepare_rows(x32, t_dev)
            segs = [e.segments() for e in engs]
            for k in range(max(len(segs[0]), len(segs[1]))):
                for br in range(2):
                    if k < len(segs[br]):
                        engs[br].run_segment(segs[br][k])
        else:
            cur = torch.cuda.current_stream(dev)
            ready = torch.cuda.Event()
            ready.record(cur)
            segs = [e.segments() for e in engs]
            for br in range(2):
                streams[br].wait_event(ready)
                with torch.cuda.stream(streams[br]):
                    engs[br].prepare_rows(x32, t_dev)
            # interleave the enqueueing segment by segment so that neither stream's host-side launches starve the other
            for k in range(max(len(segs[0]), len(segs[1]))):
                for br in range(2):
                    if k < len(segs[br]):
                        with torch.cuda.stream(streams[br]):
                            engs[br].run_segment(segs[br][k])
            for br in range(2):
                done = torch.cuda.Event()
                done.record(streams[br])
                cur.wait_event(done)
            x32.record_stream(streams[0]); x32.record_stream(streams[1])
        return _PipeHandle(pipe["out_pad"]), pipe["eps"]

"""``DiffusionDDIM`` — drop-in for the reference sampler (tools/modules/diffusions/diffusion_ddim.py:18-260,
schedules.py) registered under ``DIFFUSION``.

Same constructor keywords and the same ``ddim_sample_loop`` / ``ddim_sample`` signatures.  The fp64 schedule
tables are built once on the host exactly as the reference does (:50-68); the per-step work is different by design:
when ``model`` is the HIP ``UNetSD_T2VBase`` (possibly DDP-wrapped), one denoising step is
    [cond | uncond] UNet pass (one batched plan replay)  ->  ``vmv_cfg_ddim_step`` (CFG + x0 + DDIM update fused)
instead of two sequential forwards plus ~12 elementwise launches and a wasted ``randn_like``.
Any other callable falls back to the reference's two-call structure (needed only for API compatibility).
"""
import math

import torch

from .registry import DIFFUSION
from . import ops


def beta_schedule(schedule="cosine", num_timesteps=1000, zero_terminal_snr=False, **kwargs):
    """schedules.py:5-21 — 'linear_sd' (t2v) and 'cosine' (+ zero-terminal-SNR rescale, i2v) are the two used."""
    if schedule == "linear_sd":
        betas = torch.linspace(kwargs["init_beta"] ** 0.5, kwargs["last_beta"] ** 0.5, num_timesteps,
                               dtype=torch.float64) ** 2
    elif schedule == "linear":
        betas = torch.linspace(kwargs["init_beta"], kwargs["last_beta"], num_timesteps, dtype=torch.float64)
    elif schedule == "cosine":
        s = kwargs.get("cosine_s", 0.008)
        f = lambda u: math.cos((u + s) / (1 + s) * math.pi / 2) ** 2
        betas = torch.tensor([min(1.0 - f((i + 1) / num_timesteps) / f(i / num_timesteps), 0.999)
                              for i in range(num_timesteps)], dtype=torch.float64)
    else:
        raise NotImplementedError(f"beta schedule {schedule!r}")
    if zero_terminal_snr and betas.max() != 1.0:
        ab = (1 - betas).cumprod(0).sqrt()
        a0, aT = ab[0].clone(), ab[-1].clone()
        ab = (ab - aT) * (a0 / (a0 - aT))
        ab = ab ** 2
        betas = 1 - torch.cat([ab[0:1], ab[1:] / ab[:-1]])
    return betas


def _check_finite(t, what, is_input=False):
    """A tensor holding inf / NaN is an error that names the way out, not a video; one reduction per SAMPLE (not per step).
    Round 4: the fp16 kernels run with MODE.FP16_OVFL set (csrc/common.h) — 16-bit stores saturate at +-65504 instead of producing
    inf, and (measured, tools/experiments/nan_probe.py) the fp16 MFMA then treats a NaN operand as 0 and an inf operand as the
    largest finite value — so finite inputs and finite weights can no longer turn into inf / NaN inside the forward, and a NaN that
    ENTERS it would be swallowed by the first GEMM.  The guard therefore stands at the door: the sampler checks the noise and the
    conditioning tensors once per sample, ``decode`` / ``encode`` their input, the engines their weights once at pack time; the
    check on the RESULT stays as the belt to those braces.  ``VMV_CHECK_FINITE=0`` disables all of them."""
    import os
    if t is None or os.environ.get("VMV_CHECK_FINITE", "1") == "0" or not t.is_cuda or not t.is_floating_point():
        return
    if not bool(torch.isfinite(t).all()):
        from . import _lib as L
        if is_input:
            raise FloatingPointError(f"{what} holds inf / NaN: non-finite INPUT (the {L.elem_name()} kernels would silently drop it)")
        raise FloatingPointError(f"{what} holds inf / NaN after the {L.elem_name()} kernels: activations left the 16-bit range"
                                 + (" — rerun with hip_dtype: bf16 (VMV_DTYPE=bf16), the wide-range build" if L.elem_name() == "fp16" else ""))


def _unwrap(model):
    return getattr(model, "module", model)      # DistributedDataParallel / DataParallel


def comm_of(unet):
    return getattr(unet, "frame_comm", None)


def _kw_of_sample(kw: dict, s: int, b: int) -> dict:
    """model_kwargs of sample s of a batch of b: tensors whose leading dimension is the batch are sliced (y [b, L, D], per-sample
    camera_data [b, F, 16]), everything else (shared [1, ...] tensors, gs_data dicts, fps) is passed on as it is."""
    out = {}
    for k, v in kw.items():
        per_sample = k in ("y", "image", "local_image", "camera_data") and torch.is_tensor(v) and v.ndim >= 3 and v.shape[0] == b and b > 1
        out[k] = v[s:s + 1] if per_sample else v
    return out


@DIFFUSION.register_class()
class DiffusionDDIM(object):
    def __init__(self, schedule='linear_sd', schedule_param={}, mean_type='eps', var_type='learned_range',
                 loss_type='mse', epsilon=1e-12, rescale_timesteps=False, noise_strength=0.0, **kwargs):
        assert mean_type in ['x0', 'x_{t-1}', 'eps', 'v']
        assert var_type in ['learned', 'learned_range', 'fixed_large', 'fixed_small']
        betas = beta_schedule(schedule, **schedule_param)
        assert min(betas) > 0 and max(betas) <= 1
        self.betas = betas.double()
        self.num_timesteps = len(betas)
        self.mean_type, self.var_type, self.loss_type = mean_type, var_type, loss_type
        self.epsilon, self.rescale_timesteps, self.noise_strength = epsilon, rescale_timesteps, noise_strength
        alphas = 1 - self.betas
        self.alphas_cumprod = torch.cumprod(alphas, dim=0)
        self.alphas_cumprod_prev = torch.cat([alphas.new_ones([1]), self.alphas_cumprod[:-1]])
        self.alphas_cumprod_next = torch.cat([self.alphas_cumprod[1:], alphas.new_zeros([1])])
        self.sqrt_alphas_cumprod = torch.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = torch.sqrt(1.0 - self.alphas_cumprod)
        self.log_one_minus_alphas_cumprod = torch.log(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = torch.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = torch.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = self.betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = torch.log(self.posterior_variance.clamp(1e-20))

    def _scale_timesteps(self, t):
        if self.rescale_timesteps:
            return t.float() * 1000.0 / self.num_timesteps
        return t

    def ddim_steps(self, ddim_timesteps):
        T = self.num_timesteps
        return (1 + torch.arange(0, T, T // ddim_timesteps)).clamp(0, T - 1).flip(0)

    def step_scalars(self, step: int, stride: int):
        """fp64 table lookups cast to fp32 exactly where the reference casts them (``_i``: :9-15)."""
        f = lambda tab, i: float(tab[i].to(torch.float32))
        return dict(c_recip=f(self.sqrt_recip_alphas_cumprod, step), c_recipm1=f(self.sqrt_recipm1_alphas_cumprod, step),
                    c_sqrt_ac=f(self.sqrt_alphas_cumprod, step), c_sqrt_1mac=f(self.sqrt_one_minus_alphas_cumprod, step),
                    a_prev=f(self.alphas_cumprod, max(step - stride, 0)))

    # ------------------------------------------------------------------ fused HIP path
    def ddim_sigma(self, step: int, stride: int, eta: float) -> float:
        """sigma_t of stochastic DDIM (diffusion_ddim.py:233-236), evaluated in fp32 like the reference's ``_i`` lookups."""
        if not eta:
            return 0.0
        a = self.alphas_cumprod[step].to(torch.float32)
        ap = self.alphas_cumprod[max(step - stride, 0)].to(torch.float32)
        return float(eta * torch.sqrt((1 - ap) / (1 - a) * (1 - a / ap)))

    @staticmethod
    def _step_noise(xt, unet):
        """sigma-noise of a stochastic step (diffusion_ddim.py:239).  Frame-parallel ranks share the seed: each draws the noise of the
        WHOLE sample and keeps its own frames — independent across shards and the numbers the unsharded run draws — instead of every
        rank drawing the same local-shape block (which would repeat one noise block in every shard)."""
        comm = getattr(unet, "frame_comm", None)
        if comm is None or comm.world <= 1:
            return torch.randn_like(xt)
        b, c, fl, h, w = xt.shape
        full = torch.randn(b, c, fl * comm.world, h, w, dtype=xt.dtype, device=xt.device)
        return full[:, :, comm.rank * fl:(comm.rank + 1) * fl].contiguous()

    @torch.no_grad()
    def ddim_step_hip(self, xt, step, unet, cond_kwargs, uncond_kwargs, guide_scale, stride, x0_out=None, clamp=None, eta=0.0):
        t = torch.full((xt.shape[0],), int(step), dtype=torch.long, device=xt.device)
        eng, eps_rows = unet.forward_cfg_rows(xt, self._scale_timesteps(t), cond_kwargs, uncond_kwargs)
        sigma = self.ddim_sigma(int(step), stride, eta)
        # (the reference draws randn_like(xt) every step, eta = 0 included; only a stochastic step needs it here)
        noise = self._step_noise(xt, unet) if sigma > 0.0 else None
        k = self.step_scalars(int(step), stride)
        if xt.shape[0] == 1:
            ops.cfg_ddim_step(eps_rows, eng.out_pad, xt, float(guide_scale), v_pred=(self.mean_type == 'v'),
                              x0_out=x0_out, clamp=clamp, sigma=sigma, noise=noise, **k)
            return xt
        # b > 1 prompts in one plan: row blocks are pair-major [c_0 | u_0 | c_1 | u_1 ...] (unet_t2v._forward_cfg_rows_batched), so sample s's
        # pair starts 2 s F h w rows in and the single-sample kernel applies to it unchanged
        per = 2 * xt.shape[2] * xt.shape[3] * xt.shape[4]
        for s in range(xt.shape[0]):
            ops.cfg_ddim_step(eps_rows[s * per:(s + 1) * per], eng.out_pad, xt[s:s + 1], float(guide_scale), v_pred=(self.mean_type == 'v'),
                              x0_out=None if x0_out is None else x0_out[s:s + 1], clamp=clamp, sigma=sigma,
                              noise=None if noise is None else noise[s:s + 1], **k)
        return xt

    def _same_gs_data(self, ga, gb) -> bool:
        """Do the two CFG branches render the same views?  torch.equal on device tensors is a device-to-host sync, so the answer is
        memoised on the tensors' identity + in-place version (strong references held): once per sample, not once per refined step."""
        keys = ("input", "cam_view", "cam_view_proj")
        memo_key = tuple((id(d[k]), d[k]._version) for d in (ga, gb) for k in keys)
        memo = getattr(self, "_gs_same_memo", None)
        if memo is None or memo[0] != memo_key:
            same = all(ga[k] is gb[k] or (ga[k].shape == gb[k].shape and bool(torch.equal(ga[k], gb[k]))) for k in keys)
            self._gs_same_memo = memo = (memo_key, same, tuple(d[k] for d in (ga, gb) for k in keys))
        return memo[1]

    @torch.no_grad()
    def ddim_step_lgm(self, xt, step, unet, cond_kwargs, uncond_kwargs, guide_scale, stride, autoencoder, clamp=None, eta=0.0):
        """One LGM-refined step: each CFG branch's eps goes through predicted x0 -> 4 decoded views -> LGM Gaussians ->
        24 renders -> VAE-encoded latent_z (unet_t2v.py:404-433); CFG is applied to the two latent_z, the result is taken
        as x0 (diffusion_ddim.py:157-160,179-182), clamped like any other step's x0 (:204-205), and the DDIM update with its
        stochastic term follows from it (:233-243).  RNG order as the reference: the posterior draws of the two branches'
        encodes (inside the model calls) come BEFORE the step's sigma-noise."""
        t = torch.full((xt.shape[0],), int(step), dtype=torch.long, device=xt.device)
        eng, eps_rows = unet.forward_cfg_rows(xt, self._scale_timesteps(t), cond_kwargs, uncond_kwargs)
        k = self.step_scalars(int(step), stride)
        sigma = self.ddim_sigma(int(step), stride, eta)
        upd = lambda zc, zu: ops.ddim_x0_step(zc, zu, xt, float(guide_scale), k["c_recip"], k["c_recipm1"], k["a_prev"], clamp=clamp,
                                              sigma=sigma, noise=self._step_noise(xt, unet) if sigma > 0.0 else None)
        ref = unet.lgm_refiner(xt.device)
        comm, views, xt_all, ld = getattr(unet, "frame_comm", None), None, xt, eng.out_pad
        cfgpar = comm is not None and hasattr(comm, "exchange_branches")
        if comm is not None and (comm.world > 1 or cfgpar):
            # Frame-parallel (no reference counterpart): the branch needs x0 of 4 KEY views (frames 0 / 6 / 12 / 18), which live on
            # other ranks — one small all-gather of (x_t, eps rows) rebuilds the whole sample's on every rank (1.2 MB at 24 x 32 x 32);
            # the 4-view decode and the LGM U-Net then run replicated, and every rank renders and re-encodes ONLY its own F / R views
            # (the 24 renders + the 24-view encode are 2/3 of the branch), with the unsharded run's posterior noise for them.
            from .unet_t2v import gather_frames
            fl = xt.shape[2]
            xt_all = gather_frames(comm, xt)                                            # (CfgFrameComm: inside my branch group)
            if cfgpar:
                # CFG-parallel x frame-parallel (round 5: no longer NotImplementedError): this rank's group runs ONE branch, so it
                # runs that branch's LGM pass only — gather my branch's eps rows over the group, latent_z of my views, then the
                # partner ranks (same frames, other branch) swap their latent_z and both apply the CFG-on-latent_z update identically.
                mine = eps_rows.view(2, -1)[comm.branch].contiguous()
                allr = torch.empty(comm.world, mine.numel(), dtype=mine.dtype, device=mine.device)
                comm.all_gather(allr, mine)
                both = torch.zeros(2, allr.numel(), dtype=mine.dtype, device=mine.device)
                both[comm.branch] = allr.view(-1)
                kw_b = (cond_kwargs, uncond_kwargs)[comm.branch]
                ca, cb = (k["c_sqrt_ac"], k["c_sqrt_1mac"]) if getattr(unet, "lgm_vpred", False) else (k["c_recip"], k["c_recipm1"])
                # (host RNG: the unsharded run draws the cond branch's posterior noise, then the uncond branch's — two consecutive
                #  draws of the whole batch; every rank consumes both and uses its branch's, so the two groups' noises are
                #  independent and later draws sit at the unsharded run's RNG position)
                zb = ref.latent_z(both.view(-1, ld), ld, comm.branch, xt_all, ca, cb, autoencoder, dict(kw_b["gs_data"]),
                                  views=(comm.rank * fl, fl), rng_part=(comm.branch, 2)).contiguous()
                pair = torch.empty(2, zb.numel(), dtype=zb.dtype, device=zb.device)
                comm.exchange_branches(pair, zb.view(-1))
                upd(pair[0].view_as(zb), pair[1].view_as(zb))
                return xt
            loc = eps_rows.reshape(2, -1).contiguous()                                  # [branch][local rows x ld]
            allr = torch.empty(comm.world, loc.numel(), dtype=loc.dtype, device=loc.device)
            comm.all_gather(allr, loc)
            eps_rows = allr.view(comm.world, 2, -1).permute(1, 0, 2).reshape(-1, ld).contiguous()      # [branch][frame-major rows][ld]
            views = (comm.rank * fl, fl)
        # predicted x0 of each branch: eps form (unet_t2v.py:405) or v form (unet_i2vgen.py:441-442), following the MODEL
        ca, cb = (k["c_sqrt_ac"], k["c_sqrt_1mac"]) if getattr(unet, "lgm_vpred", False) else (k["c_recip"], k["c_recipm1"])
        ga, gb = cond_kwargs["gs_data"], uncond_kwargs["gs_data"]
        same_views = ga is gb or self._same_gs_data(ga, gb)
        if same_views and ref.pair_supported():           # both branches through every stage together (one camera set)
            z = ref.latent_z_pair(eps_rows, ld, xt_all, ca, cb, autoencoder, dict(ga), views=views)
        else:
            z = [ref.latent_z(eps_rows, ld, br, xt_all, ca, cb, autoencoder, dict(kw["gs_data"]), views=views)
                 for br, kw in enumerate((cond_kwargs, uncond_kwargs))]
        upd(z[0], z[1])
        return xt

    def _saturation_probe(self, unet, xt):
        """VMV_F16_SAT_PROBE=1 (debug, off by default): after the first step of a sample, replay the step's UNet plan launch by launch
        and warn when 16-bit GEMM outputs sit on the fp16 clamp (+-65504) — the silent failure the saturating stores make possible."""
        import os
        if os.environ.get("VMV_F16_SAT_PROBE", "0") != "1" or comm_of(unet) is not None:
            return
        engs = [e for e in getattr(unet, "_engines", {}).values() if e.B == 2 and e.F == xt.shape[2] and (e.H, e.W) == tuple(xt.shape[-2:])]
        for e in engs[:1]:
            hits = e.saturation_report()
            if hits:
                import warnings
                tot = sum(h[1] for h in hits)
                warnings.warn(f"fp16 saturation: {tot} activation values clamped at +-65504 in {len(hits)} launches (first: {hits[0][0]}, "
                              f"{hits[0][1]} of {hits[0][2]}) — this checkpoint leaves fp16's range; rerun with hip_dtype: bf16 (VMV_DTYPE=bf16)")

    @torch.no_grad()
    def ddim_sample_loop(self, noise, model, autoencoder=None, model_kwargs={}, clamp=None, percentile=None,
                         condition_fn=None, guide_scale=None, ddim_timesteps=20, eta=0.0):
        b = noise.size(0)
        steps = self.ddim_steps(ddim_timesteps)
        stride = self.num_timesteps // ddim_timesteps
        unet = _unwrap(model)
        fused = (hasattr(unet, "forward_cfg_rows") and guide_scale is not None and isinstance(model_kwargs, list)
                 and len(model_kwargs) == 2 and percentile is None
                 and condition_fn is None and self.mean_type in ('eps', 'v') and noise.is_cuda
                 and (b == 1 or (getattr(unet, "cfg_batch_ok", False)
                                 and not hasattr(getattr(unet, "frame_comm", None), "exchange_branches")
                                 and (autoencoder is None or getattr(unet, "frame_comm", None) is None))))
        # (clamp and eta > 0 ride in the fused update kernel; percentile clipping needs a quantile of the whole x0 and classifier
        #  guidance a foreign callable — both take the generic two-forward path below, as does any foreign model)
        if not fused:
            xt = noise
            for idx, step in enumerate(steps):
                t = torch.full((b,), int(step), dtype=torch.long, device=xt.device)
                ae = autoencoder if idx in (20, 30, 40) else None          # diffusion_ddim.py:254-257
                xt, _ = self.ddim_sample(xt, t, model, ae, model_kwargs, clamp, percentile, condition_fn, guide_scale,
                                         ddim_timesteps, eta)
            return xt
        assert self.var_type.startswith('fixed'), "learned variance doubles the UNet out channels: not a VideoMV config"
        comm = getattr(unet, "frame_comm", None)
        if comm is not None:   # frame-parallel: every rank denoises its own frames of the same noise; one gather at the end
            fl = noise.shape[2] // comm.world
            noise = noise[:, :, comm.rank * fl:(comm.rank + 1) * fl]
        xt = noise.detach().clone().float().contiguous()      # updated in place by the fused kernel
        kc, ku = model_kwargs
        _check_finite(xt, "the initial noise", is_input=True)           # (once per sample: see _check_finite)
        for kw in (kc, ku):
            for name, v in kw.items():
                if torch.is_tensor(v):
                    _check_finite(v, f"model_kwargs[{name!r}]", is_input=True)
        if hasattr(unet, "begin_sample"):
            unet.begin_sample()                                # new sample: step-invariant conditioning is re-evaluated
        for idx, step in enumerate(steps):
            if autoencoder is not None and idx in (20, 30, 40):      # LGM-refined steps (diffusion_ddim.py:254-256)
                if b == 1:
                    self.ddim_step_lgm(xt, int(step), unet, kc, ku, guide_scale, stride, autoencoder, clamp=clamp, eta=eta)
                else:
                    # b prompts per plan: the 47 plain steps run batched, the 3 refined ones sample by sample on the 1-prompt plan (the
                    # LGM branch decodes / renders / re-encodes per sample anyway: 2 x 24 views of 65 536 Gaussians each)
                    for s_ in range(b):
                        self.ddim_step_lgm(xt[s_:s_ + 1], int(step), unet, _kw_of_sample(kc, s_, b), _kw_of_sample(ku, s_, b), guide_scale,
                                           stride, autoencoder, clamp=clamp, eta=eta)
            else:
                self.ddim_step_hip(xt, int(step), unet, kc, ku, guide_scale, stride, clamp=clamp, eta=eta)
            if idx == 0:
                self._saturation_probe(unet, xt)
        if comm is not None:
            from .unet_t2v import gather_frames
            xt = gather_frames(comm, xt)
        _check_finite(xt, "the denoised latent")
        return xt

    # ------------------------------------------------------------------ generic path (foreign models / CPU)
    @torch.no_grad()
    def ddim_sample(self, xt, t, model, autoencoder=None, model_kwargs={}, clamp=None, percentile=None, condition_fn=None,
                    guide_scale=None, ddim_timesteps=20, eta=0.0):
        """One DDIM step for an arbitrary callable ``model`` (API compatibility; the HIP UNet never takes this
        route inside ``ddim_sample_loop``).  All samples of a call share one timestep, as in the reference loop."""
        step = int(t.reshape(-1)[0])
        ts = self._scale_timesteps(t)
        tabs = dict(autoencoder=autoencoder, sqrt_alphas_cumprod=self.sqrt_alphas_cumprod,
                    sqrt_one_minus_alphas_cumprod=self.sqrt_one_minus_alphas_cumprod,
                    sqrt_recip_alphas_cumprod=self.sqrt_recip_alphas_cumprod,
                    sqrt_recipm1_alphas_cumprod=self.sqrt_recipm1_alphas_cumprod) if autoencoder is not None else {}
        if guide_scale is None:
            pred = model(xt, ts, **model_kwargs)
        else:
            assert isinstance(model_kwargs, list) and len(model_kwargs) == 2
            cond, uncond = model(xt, ts, **tabs, **model_kwargs[0]), model(xt, ts, **tabs, **model_kwargs[1])
            pred = uncond + guide_scale * (cond - uncond)
        k = {n: torch.tensor(v, dtype=xt.dtype, device=xt.device)
             for n, v in self.step_scalars(step, self.num_timesteps // ddim_timesteps).items()}
        if autoencoder is not None:
            x0 = pred                       # the model returned latent_z (diffusion_ddim.py:179-182)
        elif self.mean_type == 'eps':
            x0 = k["c_recip"] * xt - k["c_recipm1"] * pred
        elif self.mean_type == 'v':
            x0 = k["c_sqrt_ac"] * xt - k["c_sqrt_1mac"] * pred
        elif self.mean_type == 'x0':
            x0 = pred
        else:
            raise NotImplementedError(self.mean_type)
        if percentile is not None:          # diffusion_ddim.py:200-203
            assert percentile > 0 and percentile <= 1
            sq = torch.quantile(x0.flatten(1).abs(), percentile, dim=1).clamp_(1.0).view(-1, *((1,) * (x0.ndim - 1)))
            x0 = torch.min(sq, torch.max(-sq, x0)) / sq
        elif clamp is not None:
            x0 = x0.clamp(-clamp, clamp)
        if condition_fn is not None:        # classifier guidance (:218-226): x0 -> eps, shift by the classifier's gradient, eps -> x0
            alpha = self.alphas_cumprod[step].to(xt.dtype).to(xt.device)
            eps = (k["c_recip"] * xt - x0) / k["c_recipm1"]
            kw = model_kwargs if isinstance(model_kwargs, dict) else {}
            eps = eps - (1 - alpha).sqrt() * condition_fn(xt, ts, **kw)
            x0 = k["c_recip"] * xt - k["c_recipm1"] * eps
        eps = (k["c_recip"] * xt - x0) / k["c_recipm1"]
        sigma = self.ddim_sigma(step, self.num_timesteps // ddim_timesteps, eta)
        noise = torch.randn_like(xt)        # drawn every step, as the reference does (:239), so the RNG stream matches
        mask = t.ne(0).to(xt.dtype).view(-1, *((1,) * (xt.ndim - 1)))
        direction = torch.sqrt(1 - k["a_prev"] - sigma ** 2) * eps
        return torch.sqrt(k["a_prev"]) * x0 + direction + mask * sigma * noise, x0

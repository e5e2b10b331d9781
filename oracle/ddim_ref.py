"""ORACLE (test infrastructure, CPU) — restatement of the reference DDIM sampler and beta schedules.

Follows:
  * ``beta_schedule`` / ``linear_sd_schedule`` / ``cosine_schedule`` / ``rescale_zero_terminal_snr``
        tools/modules/diffusions/schedules.py:5-21, 40-41, 50-57, 121-143
  * ``DiffusionDDIM.__init__`` tables                  tools/modules/diffusions/diffusion_ddim.py:20-68
  * ``p_mean_variance`` (CFG + eps/v/x0 -> x0)          tools/modules/diffusions/diffusion_ddim.py:138-208
  * ``ddim_sample`` / ``ddim_sample_loop``              tools/modules/diffusions/diffusion_ddim.py:210-260
Parity is pinned by ``tests/golden/schedules.safetensors`` and ``tests/golden/ddim_tiny.safetensors``
(generated from the imported reference by ``oracle/make_golden.py``) and, for the sampler options (eta > 0, clamp,
percentile, condition_fn), by ``tests/golden/ddim_options.safetensors`` (``oracle/make_golden_sampler_opts.py``).
"""
import math

import torch


def betas_for(schedule, num_timesteps=1000, init_beta=0.00085, last_beta=0.012, cosine_s=0.008,
              zero_terminal_snr=False, **_):
    if schedule == "linear_sd":
        betas = torch.linspace(init_beta ** 0.5, last_beta ** 0.5, num_timesteps, dtype=torch.float64) ** 2
    elif schedule == "cosine":
        def abar(u):
            return math.cos((u + cosine_s) / (1 + cosine_s) * math.pi / 2) ** 2
        betas = torch.tensor([min(1.0 - abar((i + 1) / num_timesteps) / abar(i / num_timesteps), 0.999)
                              for i in range(num_timesteps)], dtype=torch.float64)
    else:
        raise ValueError(schedule)
    if zero_terminal_snr and betas.max() != 1.0:
        ab_sqrt = (1 - betas).cumprod(0).sqrt()
        a0, aT = ab_sqrt[0].clone(), ab_sqrt[-1].clone()
        ab_sqrt = (ab_sqrt - aT) * (a0 / (a0 - aT))
        ab = ab_sqrt ** 2
        alphas = torch.cat([ab[0:1], ab[1:] / ab[:-1]])
        betas = 1 - alphas
    return betas


class DDIMTables:
    def __init__(self, betas):
        self.betas = betas.double()
        self.T = len(betas)
        self.ac = torch.cumprod(1 - self.betas, 0)
        self.sqrt_ac = self.ac.sqrt()
        self.sqrt_1mac = (1 - self.ac).sqrt()
        self.sqrt_recip = (1 / self.ac).sqrt()
        self.sqrt_recipm1 = (1 / self.ac - 1).sqrt()


def ddim_steps(T, ddim_timesteps):
    return (1 + torch.arange(0, T, T // ddim_timesteps)).clamp(0, T - 1).flip(0)


@torch.no_grad()
def ddim_sample_loop(noise, model, tables: DDIMTables, model_kwargs, guide_scale, ddim_timesteps=50,
                     mean_type="eps", eta=0.0, trace=None, clamp=None, percentile=None, condition_fn=None, step_noise=None):
    """``model(xt, t, **kw)`` -> eps (or v).  ``model_kwargs`` = [cond, uncond].  The sampler options follow
    diffusion_ddim.py:200-205 (percentile / clamp on x0), :218-226 (condition_fn) and :233-243 (eta: sigma, direction, noise).
    ``step_noise(i, xt)`` supplies the noise of step i (default: ``torch.randn_like`` — one draw per step, as the reference)."""
    xt = noise
    b = noise.shape[0]
    T = tables.T
    stride = T // ddim_timesteps
    for step in ddim_steps(T, ddim_timesteps):
        t = torch.full((b,), int(step), dtype=torch.long)
        y_out = model(xt, t, **model_kwargs[0])
        u_out = model(xt, t, **model_kwargs[1])
        out = u_out + guide_scale * (y_out - u_out)
        ti = int(step)

        def c(tab, i=ti):
            return tab[i].to(xt.dtype)  # fp64 table -> x dtype after lookup (diffusion_ddim.py:9-15)

        if mean_type == "eps":
            x0 = c(tables.sqrt_recip) * xt - c(tables.sqrt_recipm1) * out
        elif mean_type == "v":
            x0 = c(tables.sqrt_ac) * xt - c(tables.sqrt_1mac) * out
        else:
            raise ValueError(mean_type)
        if percentile is not None:
            sq = torch.quantile(x0.flatten(1).abs(), percentile, dim=1).clamp_(1.0).view(-1, *((1,) * (x0.ndim - 1)))
            x0 = torch.min(sq, torch.max(-sq, x0)) / sq
        elif clamp is not None:
            x0 = x0.clamp(-clamp, clamp)
        if condition_fn is not None:
            alpha = c(tables.ac)
            eps = (c(tables.sqrt_recip) * xt - x0) / c(tables.sqrt_recipm1)
            eps = eps - (1 - alpha).sqrt() * condition_fn(xt, t)
            x0 = c(tables.sqrt_recip) * xt - c(tables.sqrt_recipm1) * eps
        eps = (c(tables.sqrt_recip) * xt - x0) / c(tables.sqrt_recipm1)
        a_t = c(tables.ac)
        a_prev = tables.ac[max(ti - stride, 0)].to(xt.dtype)
        sigma = eta * torch.sqrt((1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev))
        nz = step_noise(len(trace) if trace is not None else None, xt) if step_noise is not None else torch.randn_like(xt)
        xt = torch.sqrt(a_prev) * x0 + torch.sqrt(1 - a_prev - sigma ** 2) * eps + sigma * nz
        if trace is not None:
            trace.append(xt.clone())
    return xt

"""Golden vectors for the sampler OPTIONS of the reference signature, from the IMPORTED reference (authoring container only).

    python -m oracle.make_golden_sampler_opts

``DiffusionDDIM.ddim_sample_loop(..., clamp=, percentile=, condition_fn=, eta=)`` (tools/modules/diffusions/diffusion_ddim.py:
200-205, 218-226, 233-243) — unused by the two shipped YAMLs but part of the drop-in signature (VERDICT r4 missing #4).  The model is
an analytic toy (no weights: the UNet is pinned elsewhere), the RNG is torch's CPU generator seeded per case, so a sampler that
draws one ``randn_like`` per step in the reference's order reproduces the stochastic cases bit for bit.  Writes
``tests/golden/ddim_options.safetensors`` (inputs + expected final latents); ``tests/test_sampler_options_cpu.py`` holds the product's
generic path AND the oracle restatement to it.
"""
import os

import torch
from safetensors.torch import save_file

from . import shim

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

CASES = {   # name -> sampler keywords
    "eta07": dict(eta=0.7),
    "clamp08": dict(clamp=0.8),
    "pct90": dict(percentile=0.9),
    "clamp_eta": dict(clamp=1.5, eta=0.3),
    "condfn": dict(condition_fn=True),
    "plain": dict(),
}


def toy_model(xt, t, shift=None, **_):
    """eps-predictor stand-in: smooth, depends on x_t, t and the conditioning (so CFG has two different branches)."""
    return torch.tanh(0.7 * xt + t.view(-1, 1, 1, 1, 1).float() / 1000.0) + (0.0 if shift is None else shift)


def toy_condition_fn(xt, t, **_):
    return 0.1 * torch.sin(xt) + t.view(-1, 1, 1, 1, 1).float() / 5000.0


def main():
    ns = shim.load_reference()
    D = ns.ddim.DiffusionDDIM
    out = {}
    g = torch.Generator().manual_seed(5)
    noise = torch.randn(1, 4, 3, 4, 4, generator=g)      # (b = 1: the reference's percentile path views s as [b, 1, 1, 1], which only broadcasts against a 5-D latent for b = 1)
    shift = 0.2 * torch.randn(1, 4, 3, 4, 4, generator=g)
    out["noise"], out["shift"] = noise, shift
    for mean_type in ("eps", "v"):
        dif = D(schedule="linear_sd", schedule_param=dict(num_timesteps=1000, init_beta=0.00085, last_beta=0.012,
                                                           zero_terminal_snr=False),
                mean_type=mean_type, loss_type="mse", var_type="fixed_small", rescale_timesteps=False, noise_strength=0.0)
        for name, kw in CASES.items():
            kw = dict(kw)
            if kw.pop("condition_fn", False):
                kw["condition_fn"] = toy_condition_fn
                # (the reference forwards **model_kwargs to condition_fn: a dict there, hence no CFG in this case)
                torch.manual_seed(77)
                xt = dif.ddim_sample_loop(noise=noise.clone(), model=toy_model, model_kwargs=dict(shift=shift), guide_scale=None,
                                          ddim_timesteps=10, **kw)
            else:
                torch.manual_seed(77)
                xt = dif.ddim_sample_loop(noise=noise.clone(), model=toy_model, model_kwargs=[dict(shift=shift), dict()],
                                          guide_scale=4.0, ddim_timesteps=10, **kw)
            out[f"{mean_type}.{name}"] = xt.contiguous()
            print(mean_type, name, float(xt.abs().mean()))
    save_file(out, os.path.join(GOLD, "ddim_options.safetensors"))


if __name__ == "__main__":
    main()

"""CPU tier — the sampler's host logic against fixtures written by the IMPORTED reference.

* a1: the PRODUCT's fp64 schedule tables and DDIM timesteps equal ``tests/golden/schedules.safetensors`` (VERDICT r4 #6:
  this was only ever checked by hand).
* the options of the reference signature — ``clamp``, ``percentile``, ``condition_fn``, ``eta > 0``
  (tools/modules/diffusions/diffusion_ddim.py:200-205, 218-226, 233-243) — on the product's generic path and on the oracle
  restatement, against ``tests/golden/ddim_options.safetensors`` (``oracle/make_golden_sampler_opts.py``).  The stochastic cases are
  bit-reproducible on CPU because both draw ONE ``randn_like`` per step in the same order from the same seeded generator.
"""
import os

import pytest
import torch
from safetensors.torch import load_file

from oracle.ddim_ref import DDIMTables, betas_for, ddim_sample_loop as oracle_loop
from oracle.make_golden_sampler_opts import CASES, toy_model, toy_condition_fn
from videomv_amd.diffusion_ddim import DiffusionDDIM

SCHEDULES = (("linear_sd", dict(schedule="linear_sd", schedule_param=dict(num_timesteps=1000, init_beta=0.00085, last_beta=0.012,
                                                                           zero_terminal_snr=False))),
             ("cosine_ztsnr", dict(schedule="cosine", schedule_param=dict(num_timesteps=1000, cosine_s=0.008, zero_terminal_snr=True))))


def test_product_schedule_tables_equal_the_reference(golden_dir):
    g = load_file(os.path.join(golden_dir, "schedules.safetensors"))
    for tag, kw in SCHEDULES:
        d = DiffusionDDIM(mean_type="eps", var_type="fixed_small", **kw)
        for name in ("betas", "alphas_cumprod", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
                     "sqrt_recipm1_alphas_cumprod"):
            mine, ref = getattr(d, name), g[f"{tag}.{name}"]
            assert mine.dtype == torch.float64 and mine.shape == ref.shape
            # bit-equal where finite; the zero-terminal-SNR schedule has 1 / abar_T = inf in both
            assert torch.equal(torch.isfinite(mine), torch.isfinite(ref)), (tag, name)
            fin = torch.isfinite(ref)
            assert torch.equal(mine[fin], ref[fin]), (tag, name)
        for n in (2, 20, 50):
            assert torch.equal(d.ddim_steps(n), g[f"steps{n}"])
    d = DiffusionDDIM(mean_type="eps", var_type="fixed_small", **SCHEDULES[0][1])
    assert int(d.ddim_steps(50)[0]) == 981 and int(d.ddim_steps(50)[-1]) == 1
    k = d.step_scalars(981, 20)       # fp64 lookup, then fp32 — as the reference's _i
    assert k["a_prev"] == float(g["linear_sd.alphas_cumprod"][961].to(torch.float32))
    assert k["c_recipm1"] == float(g["linear_sd.sqrt_recipm1_alphas_cumprod"][981].to(torch.float32))


def _dif(mean_type):
    return DiffusionDDIM(mean_type=mean_type, var_type="fixed_small", loss_type="mse", **SCHEDULES[0][1])


@pytest.mark.parametrize("mean_type", ["eps", "v"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_sampler_options_match_reference(golden_dir, mean_type, name):
    g = load_file(os.path.join(golden_dir, "ddim_options.safetensors"))
    kw = dict(CASES[name])
    want = g[f"{mean_type}.{name}"]
    dif = _dif(mean_type)
    tb = DDIMTables(betas_for("linear_sd"))
    if kw.pop("condition_fn", False):
        torch.manual_seed(77)
        got = dif.ddim_sample_loop(g["noise"].clone(), toy_model, model_kwargs=dict(shift=g["shift"]), guide_scale=None,
                                   ddim_timesteps=10, condition_fn=toy_condition_fn, **kw)
        assert torch.allclose(got, want, rtol=0, atol=2e-6), float((got - want).abs().max())
        return
    torch.manual_seed(77)
    got = dif.ddim_sample_loop(g["noise"].clone(), toy_model, model_kwargs=[dict(shift=g["shift"]), dict()], guide_scale=4.0,
                               ddim_timesteps=10, **kw)
    assert torch.allclose(got, want, rtol=0, atol=2e-6), float((got - want).abs().max())
    torch.manual_seed(77)
    orc = oracle_loop(g["noise"].clone(), toy_model, tb, [dict(shift=g["shift"]), dict()], guide_scale=4.0, ddim_timesteps=10,
                      mean_type=mean_type, **kw)
    assert torch.allclose(orc, want, rtol=0, atol=2e-6), float((orc - want).abs().max())


def test_sigma_is_the_reference_formula():
    d = _dif("eps")
    a, ap = d.alphas_cumprod[501].float(), d.alphas_cumprod[401].float()
    assert d.ddim_sigma(501, 100, 0.0) == 0.0
    assert d.ddim_sigma(501, 100, 0.5) == pytest.approx(float(0.5 * torch.sqrt((1 - ap) / (1 - a) * (1 - a / ap))), rel=1e-7)

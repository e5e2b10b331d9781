"""Pin the oracle (oracle/*.py, CPU fp32 restatement) against fixtures captured from the imported reference
(oracle/make_golden.py).  Tolerance: 1e-5 max-abs relative to the output scale."""
import json
import os

import pytest
import torch
from safetensors import safe_open
from safetensors.torch import load_file

from oracle.unet_ref import UNetCfg, unet_forward, block_plan
from oracle.weights import random_state_dict, unet_param_shapes, vae_decoder_param_shapes, checksum
from oracle.ddim_ref import betas_for, DDIMTables, ddim_steps, ddim_sample_loop
from oracle.vae_ref import vae_decode


def _meta(path):
    with safe_open(path, "pt") as f:
        return f.metadata()


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


@pytest.mark.parametrize("name", ["unet_tiny_a", "unet_tiny_b"])
def test_unet_forward_matches_reference(golden_dir, name):
    path = os.path.join(golden_dir, f"{name}.safetensors")
    g = load_file(path)
    meta = _meta(path)
    cfg = UNetCfg(**json.loads(meta["cfg"]))
    sd = random_state_dict(unet_param_shapes(cfg), int(meta["seed"]))
    assert checksum(sd) == pytest.approx(float(g["weights_checksum"][0]), rel=1e-12)
    taps = {}
    eps = unet_forward(sd, cfg, g["x"], g["t"], g["y"], g["camera_data"], taps=taps)
    assert eps.shape == g["eps"].shape
    assert _rel(eps, g["eps"]) < 1e-5
    # per-block taps (ResBlock incl. Cin!=Cout, SpatialTransformer, TemporalTransformer, Down/Upsample)
    inp, mid, outb = block_plan(cfg)
    names = {f"in{i}": blk[0][1] for i, blk in enumerate(inp)}
    names["mid"] = "middle_block"
    names.update({f"out{i}": blk[0][1] for i, blk in enumerate(outb)})
    n = 0
    for k, v in g.items():
        if not k.startswith("tap."):
            continue
        mine = taps[names[k[4:]]]
        assert mine.shape == v.shape, k
        assert _rel(mine, v) < 1e-5, k
        n += 1
    assert n == len(inp) + 1 + len(outb)


def test_schedules_known_answers(golden_dir):
    g = load_file(os.path.join(golden_dir, "schedules.safetensors"))
    for tag, kw in (("linear_sd", dict(schedule="linear_sd", init_beta=0.00085, last_beta=0.012)),
                    ("cosine_ztsnr", dict(schedule="cosine", cosine_s=0.008, zero_terminal_snr=True))):
        tb = DDIMTables(betas_for(**kw))
        pairs = (("betas", tb.betas), ("alphas_cumprod", tb.ac), ("sqrt_alphas_cumprod", tb.sqrt_ac),
                 ("sqrt_one_minus_alphas_cumprod", tb.sqrt_1mac), ("sqrt_recip_alphas_cumprod", tb.sqrt_recip),
                 ("sqrt_recipm1_alphas_cumprod", tb.sqrt_recipm1))
        for name, mine in pairs:
            ref = g[f"{tag}.{name}"]
            assert mine.dtype == torch.float64
            fin = torch.isfinite(ref)
            assert torch.equal(fin, torch.isfinite(mine)), (tag, name)
            assert torch.allclose(mine[fin], ref[fin], rtol=1e-12, atol=0), (tag, name)
    # SURVEY §8a known answers
    tb = DDIMTables(betas_for("linear_sd"))
    assert float(tb.ac[0]) == pytest.approx(0.99915, abs=1e-9)
    assert float(tb.ac[999]) == pytest.approx(0.0046601, abs=1e-7)
    for n in (2, 20, 50):
        assert torch.equal(ddim_steps(1000, n), g[f"steps{n}"])
    assert ddim_steps(1000, 50)[0] == 981 and ddim_steps(1000, 50)[-1] == 1


def test_ddim_two_step_cfg_matches_reference(golden_dir):
    """BASELINE config 1 analogue: 4 views, 2 DDIM steps (t = 501, 1), CFG 9, fp32 CPU."""
    g = load_file(os.path.join(golden_dir, "ddim_tiny.safetensors"))
    meta = _meta(os.path.join(golden_dir, "unet_tiny_a.safetensors"))
    cfg = UNetCfg(**json.loads(meta["cfg"]))
    sd = random_state_dict(unet_param_shapes(cfg), int(meta["seed"]))
    tb = DDIMTables(betas_for("linear_sd"))

    def model(xt, t, y, camera_data):
        return unet_forward(sd, cfg, xt, t, y, camera_data)

    kw = [dict(y=g["y"], camera_data=g["camera_data"]), dict(y=g["y_uncond"], camera_data=g["camera_data"])]
    for n in (2, 5):
        x0 = ddim_sample_loop(g["noise"].clone(), model, tb, kw, guide_scale=9.0, ddim_timesteps=n)
        assert _rel(x0, g[f"x0_steps{n}"]) < 2e-5


def test_vae_decoder_matches_reference(golden_dir):
    g = load_file(os.path.join(golden_dir, "vae_tiny.safetensors"))
    sd = random_state_dict(vae_decoder_param_shapes(ch=32), 77)
    assert checksum(sd) == pytest.approx(float(g["weights_checksum"][0]), rel=1e-12)
    img = vae_decode(sd, g["z"])
    assert _rel(img, g["img"]) < 1e-5


def test_full_size_manifest(golden_dir):
    """The oracle's parameter manifest reproduces the reference's 1484 keys / 1 412 895 300 params (F13)."""
    with open(os.path.join(golden_dir, "manifest_unet_t2v_full.json")) as f:
        man = json.load(f)
    shapes = unet_param_shapes(UNetCfg())
    assert man["n_keys"] == 1484 == len(shapes)
    assert list(shapes.keys()) == list(man["keys"].keys())
    total = 0
    for k, shp in shapes.items():
        assert list(shp) == man["keys"][k], k
        n = 1
        for d in shp:
            n *= d
        total += n
    assert total == man["n_params"] == 1412895300
    with open(os.path.join(golden_dir, "manifest_vae_full.json")) as f:
        vman = json.load(f)["keys"]
    for k, shp in vae_decoder_param_shapes().items():
        assert list(shp) == vman[k], k


def test_i2vgen_forward_matches_reference(golden_dir):
    """UNetSD_I2VGen front-end (concat x2 bug included, 64 local + 4 image context tokens, fps embedding) + trunk."""
    from oracle.unet_i2v_ref import unet_i2v_forward, i2v_param_shapes
    path = os.path.join(golden_dir, "unet_i2v_tiny.safetensors")
    g = load_file(path)
    meta = _meta(path)
    c = json.loads(meta["cfg"])
    cfg = UNetCfg(**c)
    shapes = dict(unet_param_shapes(UNetCfg(**dict(c, in_dim=8))))
    shapes.update(i2v_param_shapes(cfg))
    sd = random_state_dict({k: shapes[k] for k in sorted(shapes)}, int(meta["seed"]))
    assert checksum(sd) == pytest.approx(float(g["weights_checksum"][0]), rel=1e-12)
    out = unet_i2v_forward(sd, cfg, g["x"], g["t"], g["y"], g["image"], g["local_image"], g["fps"], g["camera_data"])
    assert _rel(out, g["out"]) < 1e-5


def test_vae_encoder_matches_reference(golden_dir):
    from oracle.vae_ref import vae_encode_moments
    from oracle.weights import vae_encoder_param_shapes
    g = load_file(os.path.join(golden_dir, "vae_enc_tiny.safetensors"))
    sd = random_state_dict(vae_encoder_param_shapes(ch=32), 91)
    assert checksum(sd) == pytest.approx(float(g["weights_checksum"][0]), rel=1e-12)
    assert _rel(vae_encode_moments(sd, g["img"]), g["moments"]) < 1e-5


def test_lgm_unet_and_gaussians_match_reference_golden(golden_dir):
    """LGM branch (the pinned part): oracle U-Net vs the imported core.unet.UNet output and two block taps,
    forward_gaussians vs core.models.LGM.forward_gaussians, get_rays vs core.utils.get_rays (all 1e-5 relative)."""
    import json
    from safetensors import safe_open
    from safetensors.torch import load_file
    from oracle.lgm_ref import LgmCfg, lgm_unet_param_shapes, lgm_unet_forward, forward_gaussians, get_rays
    from oracle.weights import random_state_dict
    path = os.path.join(golden_dir, "lgm_unet_tiny.safetensors")
    g = load_file(path)
    with safe_open(path, "pt") as f:
        meta = f.metadata()
    c = json.loads(meta["cfg"])
    cfg = LgmCfg(**{k: tuple(v) if isinstance(v, list) else v for k, v in c.items()})
    sd = random_state_dict(lgm_unet_param_shapes(cfg), int(meta["seed"]))
    taps = {}
    out = lgm_unet_forward(sd, cfg, g["x"], taps=taps)
    assert _rel(out, g["out"]) < 1e-5, _rel(out, g["out"])
    for k in ("down_blocks.2", "mid_block"):
        assert _rel(taps[k], g["tap." + k]) < 1e-5, k
    gg = load_file(os.path.join(golden_dir, "lgm_gaussians_tiny.safetensors"))
    lsd = {("unet." + k): v for k, v in sd.items()}
    lsd["conv.weight"], lsd["conv.bias"] = gg["conv.weight"], gg["conv.bias"]
    gauss = forward_gaussians(lsd, cfg, gg["images"])
    assert gauss.shape == gg["gaussians"].shape
    assert _rel(gauss, gg["gaussians"]) < 1e-5
    gr = load_file(os.path.join(golden_dir, "lgm_rays.safetensors"))
    for i in range(2):
        o, d = get_rays(gr["poses"][i], 8, 12, 39.6)
        assert torch.allclose(o, gr["rays_o"][i], atol=1e-6) and torch.allclose(d, gr["rays_d"][i], atol=1e-6)
    with open(os.path.join(golden_dir, "manifest_lgm_big.json")) as f:
        man = json.load(f)
    big = {("unet." + k): list(v) for k, v in lgm_unet_param_shapes(LgmCfg()).items()}
    big["conv.weight"], big["conv.bias"] = [14, 14, 1, 1], [14]
    assert big == man["shapes"] and man["n_params"] == 415042848

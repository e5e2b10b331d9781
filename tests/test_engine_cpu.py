"""Host-logic parity without a GPU: the recorded UNet plan (segment lists, packed weights, index maps, pooled
buffers) is executed by tests/plan_interp.py on host memory and compared with the oracle.  bf16 storage between
ops => tolerance rel-L2 <= 2e-2 on eps (the HIP kernels themselves are tested with -m gpu)."""
import ctypes
import dataclasses
import os

import pytest
import torch

from oracle.unet_ref import UNetCfg, unet_forward
from oracle.weights import random_state_dict, unet_param_shapes
from tests import plan_interp

CFG = dict(in_dim=4, dim=64, context_dim=1024, out_dim=4, dim_mult=[1, 2], num_heads=2, head_dim=64,
           num_res_blocks=1, attn_scales=[1.0, 0.5], camera_dim=16, use_camera_condition=True,
           use_fps_condition=False)


def rel_l2(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12))


def _inputs(B, F_, H, W, L, seed=5):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 4, F_, H, W, generator=g)
    t = torch.tensor([501, 21][:B])
    y = torch.randn(B, L, 1024, generator=g)
    cam = torch.randn(B, F_, 16, generator=g)
    return x, t, y, cam


def test_recorded_plan_matches_oracle(monkeypatch):
    plan_interp.install(monkeypatch)
    from videomv_amd.unet_engine import UNetEngine, param_shapes
    ocfg = UNetCfg(**{k: v for k, v in CFG.items() if k in {f.name for f in dataclasses.fields(UNetCfg)}})
    shapes = unet_param_shapes(ocfg)
    assert dict(shapes) == param_shapes(CFG)       # product manifest == oracle manifest (== reference, see golden test)
    sd = random_state_dict(shapes, 99)
    B, F_, H, W, L = 2, 3, 8, 8, 5
    x, t, y, cam = _inputs(B, F_, H, W, L)
    taps_ref = {}
    eps_ref = unet_forward(sd, ocfg, x, t, y, cam, taps=taps_ref)
    taps = {}
    eng = UNetEngine(CFG, sd, B, F_, H, W, L, torch.device("cpu"), n_t=B, taps=taps)
    eng.set_context(y)
    eng.set_camera(cam)
    eng.forward_rows(x, t)
    eps = eng.eps_ncfhw()
    # per-block taps: engine rows [(b f h w), C] vs oracle [(b f), C, h, w]
    worst = 0.0
    for key, (act, h, w) in taps.items():
        ref = taps_ref[key]
        mine = act.tensor().float().view(B * F_, h, w, act.C).permute(0, 3, 1, 2)
        assert mine.shape == ref.shape, key
        e = rel_l2(mine, ref)
        worst = max(worst, e)
        assert e < 3e-2, (key, e)
    assert eps.shape == eps_ref.shape
    assert rel_l2(eps, eps_ref) < 2e-2, rel_l2(eps, eps_ref)


def test_fused_qkv_temporal_attention_is_recorded_and_matches_oracle(monkeypatch):
    """Round 6: a TemporalTransformer of a K = 320 level records ONE launch per attention — q | k | v projection + the attention over the
    frames (VMV_EPI_TATTN, csrc/gemm_tqa.hip) with the head-major weight copies — instead of the GEMM + vmv_attention pair.  A one-level
    dim-320 network (the full model's first level, 5 heads) through the interpreter: equal to the oracle at the plan tolerance, and to
    the two-launch plan (VMV_TQA=0) to rounding."""
    from videomv_amd import _lib as L
    from videomv_amd.unet_engine import UNetEngine
    plan_interp.install(monkeypatch)
    monkeypatch.setenv("VMV_TQA_MIN_ITEMS", "1")       # (the library asks for >= 128 (row tile, head) items before it prefers the fused form)
    cfg = dict(CFG, dim=320, dim_mult=[1], num_heads=5, attn_scales=[1.0])
    ocfg = UNetCfg(**{k: v for k, v in cfg.items() if k in {f.name for f in dataclasses.fields(UNetCfg)}})
    sd = random_state_dict(unet_param_shapes(ocfg), 31)
    B, F_, H, W, Lc = 2, 6, 4, 4, 5
    x, t, y, cam = _inputs(B, F_, H, W, Lc, seed=8)
    eps_ref = unet_forward(sd, ocfg, x, t, y, cam)

    def run():
        eng = UNetEngine(cfg, sd, B, F_, H, W, Lc, torch.device("cpu"), n_t=B)
        eng.set_context(y); eng.set_camera(cam)
        eng.forward_rows(x, t)
        return eng, eng.eps_ncfhw()
    eng, eps = run()
    fused = [(lb, p_) for lb, (op, p_) in zip(eng.S.labels, eng.S.recorded) if op == L.OP_GEMM and p_.epilogue == L.EPI_TATTN]
    n_tt = sum(1 for blk in eng.inp + [eng.mid] + eng.outb for k, _, _ in blk if k == "tt")
    assert len(fused) == 2 * n_tt and all(lb.endswith("qkv+attn") for lb, _ in fused)
    assert all(p_.F == F_ and p_.P == H * W and p_.N == 960 and abs(p_.epi_scale - 0.125) < 1e-9 and p_.colsum for _, p_ in fused)
    # no temporal vmv_attention is left: the remaining attention launches are the spatial transformers' (Nq = H W)
    assert all(p_.Nq == H * W for (op, p_) in eng.S.recorded if op == L.OP_ATTENTION)
    assert rel_l2(eps, eps_ref) < 2e-2, rel_l2(eps, eps_ref)
    monkeypatch.setenv("VMV_TQA", "0")
    eng2, eps2 = run()
    assert not any(op == L.OP_GEMM and p_.epilogue == L.EPI_TATTN for op, p_ in eng2.S.recorded)
    assert eng2.S.nops >= eng.S.nops + 2 * n_tt        # (+ the LayerNorm statistics launch wherever the q | k | v GEMM is not row-stationary)
    assert rel_l2(eps, eps2) < 5e-3, rel_l2(eps, eps2)


def test_plan_replay_is_self_contained(monkeypatch):
    """the all-frame GroupNorms add into two alternating int64 accumulator buffers and each apply pass clears only the
    OTHER one, so after an odd number of such norms (or an aborted replay) the buffer norm 0 uses is dirty.  The plan's first
    launch clears both: replaying it — whole, by segments, or after garbage was left in the accumulators — needs nothing
    from prepare_rows() and gives the same eps every time."""
    monkeypatch.setenv("VMV_GN_FUSED", "0")          # tiny stat groups would otherwise take the one-launch path (no totals)
    plan_interp.install(monkeypatch)
    from videomv_amd.unet_engine import UNetEngine
    ocfg = UNetCfg(**{k: v for k, v in CFG.items() if k in {f.name for f in dataclasses.fields(UNetCfg)}})
    sd = random_state_dict(unet_param_shapes(ocfg), 99)
    B, F_, H, W, L = 2, 3, 8, 8, 5
    x, t, y, cam = _inputs(B, F_, H, W, L)
    eng = UNetEngine(CFG, sd, B, F_, H, W, L, torch.device("cpu"), n_t=B)
    assert eng.S.labels[0] == "gn.totals.clear" and eng._gn_tot_k > 0            # the totals path is really on the plan
    eng.set_context(y)
    eng.set_camera(cam)
    eng.forward_rows(x, t)
    ref = eng.eps_rows.clone()
    assert rel_l2(eng.eps_ncfhw(), unet_forward(sd, ocfg, x, t, y, cam)) < 2e-2
    for poison in (False, True):
        if poison:
            eng._gn_tot2.fill_(123456789)             # what an aborted replay could leave behind
        eng.eps_rows.zero_()
        eng.run_plan()                                # no prepare_rows()
        assert torch.equal(eng.eps_rows, ref)
    eng.eps_rows.zero_()
    for seg in eng.segments():
        eng.run_segment(seg)
    assert torch.equal(eng.eps_rows, ref)


@pytest.mark.parametrize("fold,inline", [("0", "1"), ("1", "0"), ("1", "1")])
def test_layernorm_folding_is_equivalent(monkeypatch, fold, inline):
    """VMV_FOLD_LN: LayerNorm -> Linear as one GEMM on the raw rows (packing.fold_layernorm; row statistics from a separate
    pass — rowstat — or, VMV_LN_INLINE, taken inside the GEMM's own main loop — ln_eps) or as two launches: all within the
    oracle bound, and the folded plans have no LN(x) buffer."""
    plan_interp.install(monkeypatch)
    monkeypatch.setenv("VMV_FOLD_LN", fold)
    monkeypatch.setenv("VMV_LN_INLINE", inline)
    from videomv_amd import _lib as L
    from videomv_amd.unet_engine import UNetEngine
    ocfg = UNetCfg(**{k: v for k, v in CFG.items() if k in {f.name for f in dataclasses.fields(UNetCfg)}})
    sd = random_state_dict(unet_param_shapes(ocfg), 7)
    for k in list(sd):                      # non-trivial gamma / beta so that the fold is actually exercised
        if ".norm1." in k or ".norm2." in k or ".norm3." in k:
            sd[k] = sd[k] + 0.3 * torch.randn(sd[k].shape, generator=torch.Generator().manual_seed(len(k)))
    B, F_, H, W, Lc = 2, 3, 8, 8, 5
    x, t, y, cam = _inputs(B, F_, H, W, Lc, seed=11)
    eps_ref = unet_forward(sd, ocfg, x, t, y, cam)
    eng = UNetEngine(CFG, sd, B, F_, H, W, Lc, torch.device("cpu"), n_t=B)
    eng.set_context(y)
    eng.set_camera(cam)
    eng.forward_rows(x, t)
    assert rel_l2(eng.eps_ncfhw(), eps_ref) < 2e-2
    lns = [p for op, p in eng.S.recorded if op == L.OP_LAYERNORM]
    folded = [p for op, p in eng.S.recorded if op == L.OP_GEMM and p.rowstat]
    inl = [p for op, p in eng.S.recorded if op == L.OP_GEMM and p.colsum and not p.rowstat]
    if fold == "1" and inline == "1":
        assert not lns and not folded and inl and all(p.ln_eps > 0 and p.nseg == 1 and not p.residual for p in inl)
    elif fold == "1":
        assert not inl and lns and all(p.stats_out and not p.y for p in lns) and len(folded) == len(lns)
    else:
        assert lns and all(p.y and not p.stats_out for p in lns) and not folded


def test_fold_layernorm_identity():
    """packing.fold_layernorm: rstd * (W' x - mean * colsum) + b' == W LN(x) + b in exact arithmetic (fp64 here, with the
    bf16-rounded W' on both sides)."""
    from videomv_amd import packing as P
    g = torch.Generator().manual_seed(3)
    x = torch.randn(37, 64, generator=g, dtype=torch.float64) * 1.7 + 2.5
    w, b = torch.randn(48, 64, generator=g), torch.randn(48, generator=g)
    gamma, beta = 1 + 0.2 * torch.randn(64, generator=g), 0.2 * torch.randn(64, generator=g)
    wf, bf, cs = P.fold_layernorm(w, b, gamma, beta)
    mean = x.mean(dim=1, keepdim=True)
    rstd = torch.rsqrt(x.var(dim=1, unbiased=False, keepdim=True) + 1e-5)
    got = rstd * (x @ wf.double().t() - mean * cs.double()[None, :]) + bf.double()
    w_eff = wf.double() / gamma.double()[None, :]                       # the weight the rounded W' stands for
    ref = ((x - mean) * rstd * gamma.double() + beta.double()) @ w_eff.t() + b.double() + (w.double() - w_eff) @ beta.double()
    assert torch.allclose(got, ref, rtol=1e-5, atol=1e-5)       # (b', colsum are fp32 sums)


def test_pooled_buffers_give_same_result_as_unpooled(monkeypatch):
    """Buffer recycling must not alias a live tensor: pooled run == run with recycling disabled (taps mode)."""
    plan_interp.install(monkeypatch)
    from videomv_amd.unet_engine import UNetEngine
    ocfg = UNetCfg(**{k: v for k, v in CFG.items() if k in {f.name for f in dataclasses.fields(UNetCfg)}})
    sd = random_state_dict(unet_param_shapes(ocfg), 7)
    B, F_, H, W, L = 2, 2, 4, 4, 3
    x, t, y, cam = _inputs(B, F_, H, W, L, seed=9)
    outs = []
    for taps in (None, {}):
        eng = UNetEngine(CFG, sd, B, F_, H, W, L, torch.device("cpu"), n_t=B, taps=taps)
        eng.set_context(y)
        eng.set_camera(cam)
        eng.forward_rows(x, t)
        outs.append(eng.eps_ncfhw().clone())
    assert torch.equal(outs[0], outs[1])


def test_vae_decoder_plan_matches_oracle(monkeypatch):
    """VAE decode plan (GEMM-softmax-GEMM attention, folded nin_shortcut, nearest-x2 folded into the conv gather)
    executed by the CPU interpreter vs oracle/vae_ref.py on the golden tiny configuration."""
    import json, os
    plan_interp.install(monkeypatch)
    from videomv_amd.registry import AUTO_ENCODER
    from videomv_amd.autoencoder import vae_param_shapes
    from oracle.vae_ref import vae_decode
    from oracle.weights import vae_decoder_param_shapes
    dd = dict(double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2, 4, 4],
              num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    sd = random_state_dict(vae_decoder_param_shapes(ch=32), 77)
    vae = AUTO_ENCODER.build(dict(type="AutoencoderKL", ddconfig=dd, embed_dim=4))
    missing = vae.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys
    z = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(78))
    img = vae.decode(z)
    ref = vae_decode(sd, z)
    assert img.shape == ref.shape == (2, 3, 64, 64)
    assert rel_l2(img, ref) < 2e-2, rel_l2(img, ref)
    # full-size manifest == the reference's 248 keys
    here = os.path.dirname(os.path.abspath(__file__))
    man = json.load(open(os.path.join(here, "golden", "manifest_vae_full.json")))["keys"]
    full = vae_param_shapes(dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
                                 ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[]), 4)
    assert set(full) == set(man) and all(list(full[k]) == man[k] for k in man)


def _i2v_setup(golden_dir):
    import json
    from safetensors import safe_open
    from safetensors.torch import load_file
    from oracle.unet_i2v_ref import i2v_param_shapes
    path = os.path.join(golden_dir, "unet_i2v_tiny.safetensors")
    g = load_file(path)
    with safe_open(path, "pt") as f:
        meta = f.metadata()
    c = json.loads(meta["cfg"])
    shapes = dict(unet_param_shapes(UNetCfg(**dict(c, in_dim=8))))
    shapes.update(i2v_param_shapes(UNetCfg(**c)))
    sd = random_state_dict({k: shapes[k] for k in sorted(shapes)}, int(meta["seed"]))
    return g, c, sd


def test_i2vgen_plan_matches_reference_golden(monkeypatch, golden_dir):
    """UNetSD_I2VGen (front-end kernels + trunk plan) executed by the CPU interpreter vs the eps captured from the
    imported reference; also checks the product's state-dict manifest == the reference's (655 keys at this size)."""
    plan_interp.install(monkeypatch)
    from videomv_amd.registry import MODEL
    g, c, sd = _i2v_setup(golden_dir)
    m = MODEL.build(dict(type="UNetSD_I2VGen", y_dim=1024, use_camera_condition=True, concat_dim=4, **c))
    assert set(m.state_dict().keys()) == set(sd.keys())
    m.load_state_dict(sd, strict=True)
    out = m(g["x"], g["t"], y=g["y"], image=g["image"], local_image=g["local_image"], fps=g["fps"],
            camera_data=g["camera_data"])
    assert out.shape == g["out"].shape
    assert rel_l2(out, g["out"]) < 2.5e-2, rel_l2(out, g["out"])


def test_i2vgen_two_images_per_plan_equal_the_single_image_passes(monkeypatch, golden_dir):
    """UNetSD_I2VGen, b = 2 input images of one denoising step in ONE plan of B = 4 row blocks (pair-major; round 6): every sample's
    (cond, uncond) eps rows equal the single-image fused pass of that sample (per-sample y / image / local_image, shared black image and
    negative text on the uncond branch), and slot 0 is bit-identical whatever sits in slot 1."""
    plan_interp.install(monkeypatch)
    from videomv_amd.registry import MODEL
    g, c, sd = _i2v_setup(golden_dir)
    m = MODEL.build(dict(type="UNetSD_I2VGen", y_dim=1024, use_camera_condition=True, concat_dim=4, **c)).eval()
    m.load_state_dict(sd, strict=True)
    gen = torch.Generator().manual_seed(4)
    x1 = g["x"][:1]
    _, _, F_, H, W = x1.shape
    T = F_ * H * W
    x = torch.cat([x1, torch.randn(x1.shape, generator=gen)], dim=0)
    y = torch.cat([g["y"][:1], torch.randn(g["y"][:1].shape, generator=gen)], dim=0)
    img = torch.cat([g["image"][:1], torch.randn(g["image"][:1].shape, generator=gen)], dim=0)
    li = torch.cat([g["local_image"][:1], torch.randn(g["local_image"][:1].shape, generator=gen)], dim=0)
    y0, img0 = torch.randn(g["y"][:1].shape, generator=gen), torch.zeros_like(g["image"][:1])
    fps, cam = g["fps"][:1], g["camera_data"][:1]
    t1, t2 = g["t"][:1], g["t"][:1].repeat(2)
    kw = lambda s: (dict(y=y[s:s + 1], image=img[s:s + 1], local_image=li[s:s + 1], fps=fps, camera_data=cam),
                    dict(y=y0, image=img0, local_image=li[s:s + 1], fps=fps, camera_data=cam))
    singles = [m.forward_cfg_rows(x[s:s + 1], t1, *kw(s))[1].clone() for s in range(2)]
    kc, ku = dict(y=y, image=img, local_image=li, fps=fps, camera_data=cam), dict(y=y0, image=img0, local_image=li, fps=fps, camera_data=cam)
    eng, rows = m.forward_cfg_rows(x, t2, kc, ku)
    rows = rows.clone()
    assert eng.B == 4 and rows.shape[0] == 4 * T and eng.share_prefix and eng.Bp == 2      # one camera set: the CFG prefix is shared per sample
    for s in range(2):      # (16-bit storage rounding between plans of different row counts)
        assert rel_l2(rows[2 * s * T:(2 * s + 2) * T, :4], singles[s][:, :4]) < 5e-3, s
    assert rel_l2(rows[:2 * T, :4], rows[2 * T:, :4]) > 0.05
    x_b, y_b, li_b, img_b = x.clone(), y.clone(), li.clone(), img.clone()
    x_b[1], y_b[1] = 2.0 * torch.randn(x1.shape[1:], generator=gen), torch.randn(y.shape[1:], generator=gen)
    li_b[1], img_b[1] = torch.randn(li.shape[1:], generator=gen), torch.randn(img.shape[1:], generator=gen)
    m.begin_sample()
    r_b = m.forward_cfg_rows(x_b, t2, dict(kc, y=y_b, image=img_b, local_image=li_b), dict(ku, local_image=li_b))[1]
    assert torch.equal(r_b[:2 * T], rows[:2 * T]) and rel_l2(r_b[2 * T:], rows[2 * T:]) > 0.1


def test_vae_encoder_plan_matches_reference_golden(monkeypatch, golden_dir):
    """VAE encoder plan (asymmetric-pad stride-2 convs, quant_conv, posterior sampling kernel) on the CPU interpreter vs
    the moments captured from the imported reference (non-square 64x72 image)."""
    from safetensors.torch import load_file
    plan_interp.install(monkeypatch)
    from videomv_amd.registry import AUTO_ENCODER
    from oracle.weights import vae_encoder_param_shapes
    from oracle.vae_ref import posterior_sample
    g = load_file(os.path.join(golden_dir, "vae_enc_tiny.safetensors"))
    dd = dict(double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2, 4, 4],
              num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    sd = random_state_dict(vae_encoder_param_shapes(ch=32), 91)
    vae = AUTO_ENCODER.build(dict(type="AutoencoderKL", ddconfig=dd, embed_dim=4))
    vae.load_state_dict(sd, strict=False)
    post = vae.encode(g["img"])
    mom = post.parameters
    assert mom.shape == g["moments"].shape
    assert rel_l2(mom, g["moments"]) < 2e-2, rel_l2(mom, g["moments"])
    torch.manual_seed(5)
    z = vae.encode_firsr_stage(g["img"], 0.18215)
    torch.manual_seed(5)
    noise = torch.randn(2, 4, 8, 9)
    assert rel_l2(z, posterior_sample(g["moments"], noise, 0.18215)) < 2e-2


def test_lgm_plan_matches_oracle_and_reference_golden(monkeypatch, golden_dir):
    """SURVEY a16 (the pinned part): the recorded LGM plan (ResnetBlocks with folded skip_scale, MVAttention on the flash
    path at head_dim 64 and on the GEMM/softmax/GEMM path at head_dim 32, Gaussian activations incl. the dim=1 normalize)
    executed by the interpreter vs the oracle taps and the imported reference's Gaussians.  bf16 storage: rel-L2 <= 3e-2
    on the taps / raw features; activated Gaussians <= 2e-2."""
    import json
    from safetensors import safe_open
    from safetensors.torch import load_file
    plan_interp.install(monkeypatch)
    from videomv_amd.lgm import LgmEngine, LgmOptions, lgm_param_shapes
    from oracle.lgm_ref import LgmCfg, lgm_unet_param_shapes, lgm_unet_forward
    path = os.path.join(golden_dir, "lgm_unet_tiny.safetensors")
    with safe_open(path, "pt") as f:
        meta = f.metadata()
    c = {k: tuple(v) if isinstance(v, list) else v for k, v in json.loads(meta["cfg"]).items()}
    gg = load_file(os.path.join(golden_dir, "lgm_gaussians_tiny.safetensors"))
    cfg = LgmCfg(**c)
    opt = LgmOptions(**c, input_size=32, splat_size=32, output_size=64)
    sd = random_state_dict(lgm_unet_param_shapes(cfg), int(meta["seed"]))
    lsd = {("unet." + k): v for k, v in sd.items()}
    lsd["conv.weight"], lsd["conv.bias"] = gg["conv.weight"], gg["conv.bias"]
    assert {k: tuple(v.shape) for k, v in lsd.items()} == lgm_param_shapes(opt)
    images = gg["images"][0]
    taps_ref = {}
    lgm_unet_forward(sd, cfg, images, taps=taps_ref)
    taps = {}
    eng = LgmEngine(opt, lsd, 32, 32, torch.device("cpu"), taps=taps)
    gauss = eng.forward_gaussians(images)
    for key, (act, h, w) in taps.items():
        mine = act.tensor().float().view(4, h, w, act.C).permute(0, 3, 1, 2)
        assert mine.shape == taps_ref[key].shape, key
        assert rel_l2(mine, taps_ref[key]) < 3e-2, (key, rel_l2(mine, taps_ref[key]))
    assert gauss.shape == gg["gaussians"][0].shape
    assert rel_l2(gauss, gg["gaussians"][0]) < 2e-2, rel_l2(gauss, gg["gaussians"][0])


def test_fps_condition_matches_oracle(monkeypatch):
    """UNetSD_T2VBase(use_fps_condition=True): fps_embedding(sinusoidal(fps)) is added to the time embedding
    (unet_t2v.py:155-161,323-324).  Registry-built model through the interpreter vs the oracle; fps=None leaves it out."""
    plan_interp.install(monkeypatch)
    from videomv_amd.registry import MODEL
    import videomv_amd.unet_t2v  # noqa: F401
    cfg = dict(CFG, use_fps_condition=True)
    ocfg = UNetCfg(**{k: v for k, v in cfg.items() if k in {f.name for f in dataclasses.fields(UNetCfg)}})
    shapes = dict(unet_param_shapes(ocfg))
    E = 4 * cfg["dim"]
    shapes.update({"fps_embedding.0.weight": (E, cfg["dim"]), "fps_embedding.0.bias": (E,),
                   "fps_embedding.2.weight": (E, E), "fps_embedding.2.bias": (E,)})
    sd = random_state_dict({k: shapes[k] for k in sorted(shapes)}, 17)
    m = MODEL.build(dict(type="UNetSD_T2VBase", y_dim=1024, **{k: v for k, v in cfg.items()})).eval()
    m.load_state_dict(sd, strict=True)
    B, F_, H, W, Lc = 1, 3, 8, 8, 5
    x, t, y, cam = _inputs(B, F_, H, W, Lc, seed=23)
    fps = torch.tensor([8])
    e_fps = m(x, t, y=y, camera_data=cam, fps=fps)
    e_none = m(x, t, y=y, camera_data=cam, fps=None)
    r_fps = unet_forward(sd, ocfg, x, t, y, cam, fps=fps)
    r_none = unet_forward(sd, ocfg, x, t, y, cam, fps=None)
    assert rel_l2(e_fps, r_fps) < 1e-2 and rel_l2(e_none, r_none) < 1e-2
    assert rel_l2(e_fps, r_none) > 5 * rel_l2(e_fps, r_fps)          # the fps term is actually in there


def test_cfg_conditioning_cache_follows_the_tensors(monkeypatch):
    """The fused CFG pass evaluates the step-invariant conditioning once per sample.  The cache must notice (a) an
    in-place refill of the same `y` buffer, (b) new tensors (possibly at a recycled address) and (c) be dropped by
    ddim_sample_loop at the start of every sample; unchanged tensors must hit (no re-evaluation)."""
    plan_interp.install(monkeypatch)
    from videomv_amd.registry import MODEL
    import videomv_amd.unet_t2v  # noqa: F401
    ocfg = UNetCfg(**{k: v for k, v in CFG.items() if k in {f.name for f in dataclasses.fields(UNetCfg)}})
    sd = random_state_dict(unet_param_shapes(ocfg), 3)
    m = MODEL.build(dict(type="UNetSD_T2VBase", y_dim=1024, **CFG)).eval()
    m.load_state_dict(sd, strict=True)
    F_, H, W, Lc = 2, 8, 8, 5
    g = torch.Generator().manual_seed(1)
    xt = torch.randn(1, 4, F_, H, W, generator=g)
    t = torch.tensor([501])
    ya, yb, y0 = (torch.randn(1, Lc, 1024, generator=g) for _ in range(3))
    cam = torch.randn(1, F_, 16, generator=g)

    def run(y):
        eng, rows = m.forward_cfg_rows(xt, t, dict(y=y, camera_data=cam), dict(y=y0, camera_data=cam))
        return eng, rows.clone()

    eng, ra = run(ya)
    calls = []
    orig = eng.set_context
    monkeypatch.setattr(eng, "set_context", lambda y: (calls.append(1), orig(y))[1])
    _, ra2 = run(ya)
    assert not calls and torch.equal(ra, ra2)                         # hit: nothing re-evaluated
    buf = ya.clone()
    _, r1 = run(buf)
    assert len(calls) == 1 and torch.equal(r1, ra)                    # new tensor object: re-evaluated
    buf.copy_(yb)                                                     # in-place refill of the SAME tensor
    _, r2 = run(buf)
    assert len(calls) == 2 and not torch.equal(r2, ra)
    _, rb = run(yb)
    assert torch.equal(r2, rb)
    n = len(calls)
    m.begin_sample()                                                  # what ddim_sample_loop does per sample
    _, rb2 = run(yb)
    assert len(calls) == n + 1 and torch.equal(rb, rb2)
    # differing camera_data on the two branches is honoured (per-branch rows), not silently dropped
    cam_u = torch.randn(1, F_, 16, generator=g)
    _, rc = m.forward_cfg_rows(xt, t, dict(y=ya, camera_data=cam), dict(y=y0, camera_data=cam_u))
    T = F_ * H * W
    e_u = m(xt, t, y=y0, camera_data=cam_u)                           # reference-structured single-branch forward
    mine_u = rc[T:, :4].reshape(F_, H * W, 4).permute(2, 0, 1).reshape(1, 4, F_, H, W)
    assert rel_l2(mine_u, e_u) < 2e-3


def test_batched_prompts_in_one_cfg_plan_equal_the_single_prompt_passes(monkeypatch):
    """b = 2 prompts of one denoising step in ONE plan of B = 4 row blocks, pair-major [c_0 | u_0 | c_1 | u_1] (round 6: the small
    levels of one sample do not fill the chip; the reference's sampler API admits noise [b, 4, F, h, w], diffusion_ddim.py:247-260).
    Every sample's eps rows and its fused CFG + DDIM update must equal the single-prompt pass of that sample: shared uncond text and
    camera ([1, ...]), per-sample cond text; then per-sample cameras on both branches."""
    plan_interp.install(monkeypatch)
    from videomv_amd.registry import MODEL, DIFFUSION
    import videomv_amd.unet_t2v  # noqa: F401
    import videomv_amd.diffusion_ddim  # noqa: F401
    ocfg = UNetCfg(**{k: v for k, v in CFG.items() if k in {f.name for f in dataclasses.fields(UNetCfg)}})
    sd = random_state_dict(unet_param_shapes(ocfg), 5)
    m = MODEL.build(dict(type="UNetSD_T2VBase", y_dim=1024, **CFG)).eval()
    m.load_state_dict(sd, strict=True)
    F_, H, W, Lc = 2, 8, 8, 5
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 4, F_, H, W, generator=g)
    t1, t2 = torch.tensor([501]), torch.tensor([501, 501])
    yc = torch.randn(2, Lc, 1024, generator=g)
    y0 = torch.randn(1, Lc, 1024, generator=g)
    cam = torch.randn(1, F_, 16, generator=g)
    T = F_ * H * W
    singles = [m.forward_cfg_rows(x[s:s + 1], t1, dict(y=yc[s:s + 1], camera_data=cam), dict(y=y0, camera_data=cam))[1].clone() for s in range(2)]
    eng, rows = m.forward_cfg_rows(x, t2, dict(y=yc, camera_data=cam), dict(y=y0, camera_data=cam))
    assert eng.B == 4 and rows.shape[0] == 4 * T
    # one camera set => the CFG prefix is recorded on ONE row block per prompt (2 of the 4) and replicated pairwise, as in the 1-prompt pass
    from videomv_amd import _lib as L
    assert eng.share_prefix and eng.Bp == 2
    gm = [p.M for op, p in eng.S.recorded if op == L.OP_GEMM]
    assert gm.count(2 * T) >= 10 and sum(1 for l in eng.S.labels if ".share." in l or l.startswith("share.")) == 3
    to_ncfhw = lambda r: r[:, :4].reshape(F_, H * W, 4).permute(2, 0, 1).reshape(1, 4, F_, H, W)
    for s in range(2):
        # (16-bit storage rounding: the single-prompt pass records its CFG prefix once on one branch's rows, the B = 4 plan does not)
        assert rel_l2(rows[2 * s * T:(2 * s + 2) * T, :4], singles[s][:, :4]) < 3e-3, s
        for br, yy in enumerate((yc[s:s + 1], y0)):                          # and each row block against the fp32 oracle of its own (x, y)
            ref = unet_forward(sd, ocfg, x[s:s + 1], t1, yy, cam)
            blk = rows[(2 * s + br) * T:(2 * s + br + 1) * T]
            assert rel_l2(to_ncfhw(blk), ref) < 1e-2, (s, br)
    assert rel_l2(rows[:2 * T, :4], rows[2 * T:, :4]) > 0.05                 # (the two prompts do differ)
    # the whole fused step, per sample on its row offset: against the reference-structured update (DiffusionDDIM.ddim_sample) fed the SAME
    # eps rows (guidance 9 would amplify the storage-rounding difference between two plans: compare the update, not the plans, here)
    dif = DIFFUSION.build(dict(type="DiffusionDDIM", schedule="linear_sd", schedule_param=dict(num_timesteps=1000, init_beta=0.00085, last_beta=0.0120),
                               mean_type="eps", var_type="fixed_small"))
    kc, ku = dict(y=yc, camera_data=cam), dict(y=y0, camera_data=cam)
    xb, x0b = x.clone(), torch.empty_like(x)
    dif.ddim_step_hip(xb, 501, m, kc, ku, 9.0, 500, x0_out=x0b, clamp=1.5)
    for s in range(2):
        fake = lambda xt, ts, br=0, s=s: to_ncfhw(rows[(2 * s + br) * T:(2 * s + br + 1) * T])
        want, want0 = dif.ddim_sample(x[s:s + 1], torch.tensor([501]), fake, None, [dict(br=0), dict(br=1)], 1.5, None, None, 9.0, 2, 0.0)
        assert rel_l2(xb[s:s + 1], want) < 1e-5 and rel_l2(x0b[s:s + 1], want0) < 1e-5, s
    assert float(x0b.abs().max()) <= 1.5 and rel_l2(xb[:1], xb[1:]) > 0.05
    # stochastic step: sample s gets ITS slice of the one randn_like(x_t) the batch draws
    xe, x0e = x.clone(), torch.empty_like(x)
    torch.manual_seed(3)
    dif.ddim_step_hip(xe, 501, m, kc, ku, 9.0, 500, x0_out=x0e, clamp=1.5, eta=0.7)
    torch.manual_seed(3)
    noise = torch.randn_like(x)
    k = dif.step_scalars(501, 500)
    sg = dif.ddim_sigma(501, 500, 0.7)
    eps_hat = (k["c_recip"] * x - x0e) / k["c_recipm1"]
    want = (k["a_prev"] ** 0.5) * x0e + ((1 - k["a_prev"] - sg * sg) ** 0.5) * eps_hat + sg * noise
    assert torch.equal(x0e, x0b) and rel_l2(xe, want) < 1e-5 and sg > 0
    # per-sample cameras, different on the two branches
    cams_c, cams_u = torch.randn(2, F_, 16, generator=g), torch.randn(2, F_, 16, generator=g)
    eng_c, rows_c = m.forward_cfg_rows(x, t2, dict(y=yc, camera_data=cams_c), dict(y=y0, camera_data=cams_u))
    assert not eng_c.share_prefix and eng_c is not eng
    for s in range(2):
        one = m.forward_cfg_rows(x[s:s + 1], t1, dict(y=yc[s:s + 1], camera_data=cams_c[s:s + 1]), dict(y=y0, camera_data=cams_u[s:s + 1]))[1]
        assert rel_l2(rows_c[2 * s * T:(2 * s + 2) * T, :4], one[:, :4]) < 3e-3, s
    # samples never see each other (all-frame GroupNorm statistics, attention batches, context K / V are per row block): another prompt
    # and latent in slot 1 leaves slot 0's rows BIT-identical
    r_a = m.forward_cfg_rows(x, t2, dict(y=yc, camera_data=cam), dict(y=y0, camera_data=cam))[1].clone()
    x_b, y_b = x.clone(), yc.clone()
    x_b[1], y_b[1] = 3.0 * torch.randn(4, F_, H, W, generator=g), torch.randn(Lc, 1024, generator=g)
    r_b = m.forward_cfg_rows(x_b, t2, dict(y=y_b, camera_data=cam), dict(y=y0, camera_data=cam))[1]
    assert torch.equal(r_a[:2 * T], r_b[:2 * T]) and rel_l2(r_a[2 * T:], r_b[2 * T:]) > 0.1
    with pytest.raises(ValueError):
        m.forward_cfg_rows(x, t2, dict(y=torch.randn(3, Lc, 1024), camera_data=cam), dict(y=y0, camera_data=cam))


def test_shared_cfg_prefix_and_hoisted_context_kv(monkeypatch):
    """B = 2 CFG pair: (a) everything before the first cross-attention is recorded once on one branch's rows and
    replicated (share_prefix) — same eps as the plain B = 2 plan and as the oracle, fewer rows in the prefix GEMMs;
    (b) K / V of the text tokens are not in the per-step plan at all: they sit in the context stream that set_context
    runs once per sample (one GEMM on B*L rows per cross-attention layer)."""
    plan_interp.install(monkeypatch)
    from videomv_amd import _lib as L
    from videomv_amd.unet_engine import UNetEngine
    ocfg = UNetCfg(**{k: v for k, v in CFG.items() if k in {f.name for f in dataclasses.fields(UNetCfg)}})
    sd = random_state_dict(unet_param_shapes(ocfg), 41)
    B, F_, H, W, Lc = 2, 3, 8, 8, 5
    g = torch.Generator().manual_seed(8)
    x1 = torch.randn(1, 4, F_, H, W, generator=g)
    y = torch.randn(B, Lc, 1024, generator=g)
    cam = torch.randn(1, F_, 16, generator=g)
    t = torch.tensor([501])
    out = {}
    for share in (False, True):
        eng = UNetEngine(CFG, sd, B, F_, H, W, Lc, torch.device("cpu"), n_t=1, share_prefix=share)
        eng.set_context(y)
        eng.set_camera(cam)
        eng.forward_rows(x1, t)
        out[share] = (eng, eng.eps_ncfhw().clone())
    e0, e1 = out[False][1], out[True][1]
    ref = torch.cat([unet_forward(sd, ocfg, x1, t, y[b:b + 1], cam) for b in range(B)], dim=0)
    assert rel_l2(e0, ref) < 1e-2 and rel_l2(e1, ref) < 1e-2
    assert rel_l2(e1, e0) < 1e-3, rel_l2(e1, e0)
    for share in (False, True):
        eng = out[share][0]
        labels = eng.S.labels
        assert not any(l.endswith(".kv") for l in labels)                       # no context K/V GEMM per step
        n_cross = sum(1 for blk in eng.inp + [eng.mid] + eng.outb for k, _, _ in blk if k == "st")
        assert eng.Sctx.nops == n_cross and all(l.endswith(".kv") for l in eng.Sctx.labels)
        assert all(p.M == B * Lc for _, p in eng.Sctx.recorded)
    T = B * F_ * H * W
    gm0 = [p.M for op, p in out[False][0].S.recorded if op == L.OP_GEMM]
    gm1 = [p.M for op, p in out[True][0].S.recorded if op == L.OP_GEMM]
    assert gm0.count(T // 2) == 0 and gm1.count(T // 2) >= 10                   # prefix GEMMs run on one branch's rows
    assert sum(1 for l in out[True][0].S.labels if ".share." in l or l.startswith("share.")) == 3
    # a context refill is picked up (K / V recomputed by set_context)
    eng = out[True][0]
    y2 = torch.randn(B, Lc, 1024, generator=g)
    eng.set_context(y2)
    eng.forward_rows(x1, t)
    ref2 = torch.cat([unet_forward(sd, ocfg, x1, t, y2[b:b + 1], cam) for b in range(B)], dim=0)
    assert rel_l2(eng.eps_ncfhw(), ref2) < 1e-2
    with pytest.raises(ValueError):
        eng.set_camera(torch.randn(B, F_, 16, generator=g))                    # per-branch cameras cannot share a prefix


def test_vae_engines_share_packed_weights(monkeypatch):
    """decode / encode engines of one AutoencoderKL for other frame counts share ONE packed weight copy, and a
    load_state_dict drops them all; results are unchanged."""
    plan_interp.install(monkeypatch)
    from videomv_amd.registry import AUTO_ENCODER
    import videomv_amd.autoencoder  # noqa: F401
    from oracle.weights import random_state_dict, vae_decoder_param_shapes
    from oracle.vae_ref import vae_decode
    dd = dict(double_z=True, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2, 4, 4],
              num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    sd = random_state_dict(vae_decoder_param_shapes(ch=32), 77)
    vae = AUTO_ENCODER.build(dict(type="AutoencoderKL", ddconfig=dd, embed_dim=4))
    vae.load_state_dict(sd, strict=False)
    z = torch.randn(2, 4, 4, 4, generator=torch.Generator().manual_seed(3))
    img2 = vae.decode(z)
    img1 = vae.decode(z[:1])
    engs = list(vae._engines.values())
    assert len(engs) == 2 and engs[0].wt is engs[1].wt
    ref = vae_decode(sd, z)
    assert rel_l2(img2, ref) < 2e-2 and rel_l2(img1, ref[:1]) < 2e-2
    vae.load_state_dict(sd, strict=False)
    assert not vae._engines


def test_lgm_plan_two_samples_equals_two_plans(monkeypatch, golden_dir):
    """LgmEngine(batch=2) — the plan the batched LGM branch runs for both CFG branches at once — against two single-sample
    plans on the same weights (shared packed copy): per-image stages see twice the rows, the multi-view attention runs once
    per sample (n_outer = 2 with a per-sample token stride)."""
    import json
    from safetensors import safe_open
    from safetensors.torch import load_file
    plan_interp.install(monkeypatch)
    from videomv_amd.lgm import LgmEngine, LgmOptions
    from oracle.lgm_ref import LgmCfg, lgm_unet_param_shapes
    path = os.path.join(golden_dir, "lgm_unet_tiny.safetensors")
    with safe_open(path, "pt") as f:
        meta = f.metadata()
    c = {k: tuple(v) if isinstance(v, list) else v for k, v in json.loads(meta["cfg"]).items()}
    gg = load_file(os.path.join(golden_dir, "lgm_gaussians_tiny.safetensors"))
    opt = LgmOptions(**c, input_size=32, splat_size=32, output_size=64)
    sd = random_state_dict(lgm_unet_param_shapes(LgmCfg(**c)), int(meta["seed"]))
    lsd = {("unet." + k): v for k, v in sd.items()}
    lsd["conv.weight"], lsd["conv.bias"] = gg["conv.weight"], gg["conv.bias"]
    a = gg["images"][0]
    b = a.flip(0) * 0.7 + 0.1 * torch.randn(a.shape, generator=torch.Generator().manual_seed(5))
    one = LgmEngine(opt, lsd, 32, 32, torch.device("cpu"))
    ga = one.forward_gaussians(a).clone()
    gb = one.forward_gaussians(b).clone()
    two = LgmEngine(opt, lsd, 32, 32, torch.device("cpu"), batch=2, packed=one.wt)
    assert two.wt is one.wt
    g2 = two.forward_gaussians(torch.cat([a, b])).view(2, -1, 14).clone()
    assert g2.shape[1:] == ga.shape
    # same arithmetic per image; the doubled row count changes split-K factors / accumulation order, i.e. isolated 16-bit
    # roundings that a whole U-Net carries along: held to the per-block bound
    assert rel_l2(g2[0], ga) < 1e-2 and rel_l2(g2[1], gb) < 1e-2, (rel_l2(g2[0], ga), rel_l2(g2[1], gb))
    assert rel_l2(ga, gb) > 0.05          # (the two samples do differ)
    # independence: another second sample leaves the first sample's Gaussians unchanged (the attention never mixes samples;
    # up to the host BLAS's position-dependent blocking, orders of magnitude below the effect of a mixed-in sample)
    g3 = two.forward_gaussians(torch.cat([a, a])).view(2, -1, 14)
    assert rel_l2(g3[0], g2[0]) < 1e-3 and rel_l2(g3[1], g3[0]) < 1e-3, (rel_l2(g3[0], g2[0]), rel_l2(g3[1], g3[0]))


@pytest.mark.parametrize("batched", ["1", "0"])
def test_vae_attention_grouped_weights_matches_oracle(monkeypatch, batched):
    """The VAE's single-head attention with all frames in one launch per stage (grouped weights: VmvGemmParams.wgroup_rows)
    and per frame (VMV_VAE_ATTN_BATCHED=0): a decoder whose mid block has 256 channels at 16 x 16 (h w = 256: a 256-row tile
    never straddles two frames) through the interpreter against the fp32 oracle decoder."""
    plan_interp.install(monkeypatch)
    monkeypatch.setenv("VMV_VAE_ATTN_BATCHED", batched)
    from videomv_amd.registry import AUTO_ENCODER
    from videomv_amd import _lib as L
    import videomv_amd.autoencoder  # noqa: F401
    from oracle.weights import random_state_dict, vae_decoder_param_shapes
    from oracle.vae_ref import vae_decode
    dd = dict(double_z=True, z_channels=4, resolution=128, in_channels=3, out_ch=3, ch=64, ch_mult=[1, 2, 4, 4],
              num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    sd = random_state_dict(vae_decoder_param_shapes(ch=64), 78)
    vae = AUTO_ENCODER.build(dict(type="AutoencoderKL", ddconfig=dd, embed_dim=4))
    vae.load_state_dict(sd, strict=False)
    z = torch.randn(3, 4, 16, 16, generator=torch.Generator().manual_seed(4))
    img = vae.decode(z)
    eng = next(iter(vae._engines.values()))
    grouped = [p for op, p in eng.S.recorded if op == L.OP_GEMM and p.wgroup_rows > 0]
    if batched == "1":
        assert len(grouped) == 3 and {p.wgroup_rows for p in grouped} == {256}          # V^T, Q K^T, P V of all 3 frames
    else:
        assert not grouped
    assert rel_l2(img, vae_decode(sd, z)) < 2e-2


def test_tile_table_hook_and_rerecord(monkeypatch):
    """ops.make_tuner / UNetEngine._rerecord (host logic of the measured tile table): an entry for a launch's signature
    forces its tile and split-K factor (with a workspace) when the plan is recorded; launches without an entry stay on the policy;
    recording the plan again after the table changed (what VMV_AUTOTUNE=1 does) gives the same result as a fresh engine."""
    plan_interp.install(monkeypatch)
    from videomv_amd import _lib as L, ops
    from videomv_amd.unet_engine import UNetEngine
    ocfg = UNetCfg(**{k: v for k, v in CFG.items() if k in {f.name for f in dataclasses.fields(UNetCfg)}})
    sd = random_state_dict(unet_param_shapes(ocfg), 99)
    B, F_, H, W, Lc = 2, 3, 8, 8, 5
    x, t, y, cam = _inputs(B, F_, H, W, Lc)
    monkeypatch.setenv("VMV_TILE_RULES", "0")           # (the table hook alone: the default rule would also force tiles on this tiny net)
    monkeypatch.setattr(ops, "_TUNED", {})
    eng = UNetEngine(CFG, sd, B, F_, H, W, Lc, torch.device("cpu"), n_t=B)
    assert eng.n_tuned == 0
    eng.set_context(y); eng.set_camera(cam); eng.forward_rows(x, t)
    eps0 = eng.eps_ncfhw().clone()
    gemms = [(lb, p) for lb, (op, p) in zip(eng.S.labels, eng.S.recorded) if op == L.OP_GEMM and p.tile == L.TILE_AUTO and not p.rowstat and p.ln_eps == 0]
    lb, p = next((lb, p) for lb, p in gemms if lb.endswith(".conv1"))
    sig = ops.gemm_signature(p)
    n_same = sum(1 for _, q in gemms if ops.gemm_signature(q) == sig)
    monkeypatch.setattr(ops, "_TUNED", {sig: dict(tile=L.TILE_G128x128, ksplit=2)})
    eng._rerecord()                                    # what VMV_AUTOTUNE=1 does after measuring
    assert eng.n_tuned == n_same >= 1
    forced = [q for op, q in eng.S.recorded if op == L.OP_GEMM and ops.gemm_signature(q) == sig]
    assert all(q.tile == L.TILE_G128x128 and q.ksplit == 2 and q.workspace for q in forced)
    assert all(q.tile == L.TILE_AUTO for op, q in eng.S.recorded if op == L.OP_GEMM and ops.gemm_signature(q) != sig)
    eng.set_context(y); eng.set_camera(cam); eng.forward_rows(x, t)
    assert torch.equal(eng.eps_ncfhw(), eps0)          # (the interpreter ignores tile / split-K: same arithmetic, same bits)
    fresh = UNetEngine(CFG, sd, B, F_, H, W, Lc, torch.device("cpu"), n_t=B)
    assert fresh.n_tuned == n_same and fresh.S.nops == eng.S.nops


def test_tuned_table_matches_the_full_size_plans(monkeypatch):
    """videomv_amd/tuned_gemm.json is keyed by launch signature: if the engine's recording changes (another segment order, a new flag)
    the table silently stops applying.  The full-size plans are recorded here on the CPU (zero weights, seconds) and every entry tagged
    with a world-1 / simulated-rank plan must still name a launch of that plan; the counts pin how much of each plan is tuned."""
    import json
    plan_interp.install(monkeypatch)
    from videomv_amd import _lib as L, ops
    from videomv_amd.comm import SimComm
    from videomv_amd.unet_engine import UNetEngine, param_shapes
    monkeypatch.setattr(ops, "_TUNED", None)
    monkeypatch.delenv("VMV_TUNED", raising=False)
    monkeypatch.delenv("VMV_TUNED_FILE", raising=False)
    cfg = dict(in_dim=4, dim=320, context_dim=1024, out_dim=4, dim_mult=[1, 2, 4, 4], num_heads=8, head_dim=64, num_res_blocks=2,
               attn_scales=[1.0, 0.5, 0.25], camera_dim=16, use_camera_condition=True, use_fps_condition=False)
    sd = {k: torch.zeros(s) for k, s in param_shapes(cfg).items()}
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "videomv_amd", "tuned_gemm.json")) as f:
        tab = json.load(f)["fp16"]
    dev = torch.device("cpu")
    e64 = UNetEngine(cfg, sd, 2, 24, 40, 64, 77, dev, n_t=1, share_prefix=True)
    e32 = UNetEngine(cfg, sd, 2, 24, 32, 32, 77, dev, n_t=1, share_prefix=True, packed=e64.packed)
    e8 = UNetEngine(cfg, sd, 2, 24, 40, 64, 77, dev, n_t=1, comm=SimComm(8, 0), packed=e64.packed)
    sigs = lambda e: {ops.gemm_signature(p) for op, p in e.S.recorded if op == L.OP_GEMM}
    # (round 5: the fill rule is the default and the table holds only what it does not reproduce — n_ruled counts the rule's launches)
    tot = lambda e: e.n_tuned + getattr(e, "n_ruled", 0)
    assert e64.S.nops == 766 and tot(e64) >= 80 and tot(e32) >= 200 and tot(e8) >= 150, [(e.n_tuned, getattr(e, "n_ruled", 0)) for e in (e64, e32, e8)]
    assert all(getattr(e, "n_stale", 0) == 0 for e in (e64, e32, e8))       # no entry of the packaged table is refused by the library
    for tag, eng in (("world1 40x64", e64), ("world1 32x32", e32), ("world8 rank0 B=2 40x64", e8)):
        mine = {k for k, v in tab.items() if v["plan"] == tag}
        assert mine and mine <= sigs(eng), (tag, sorted(mine - sigs(eng))[:3])
    # every forced choice is one the library accepts for that launch (host-side validation: no launch)
    for eng in (e64, e32, e8):
        for op, p in eng.S.recorded:
            if op == L.OP_GEMM and p.tile != L.TILE_AUTO:
                assert eng.S.lib.vmv_gemm_pick_tile(ctypes.byref(p)) == p.tile
                assert eng.S.lib.vmv_gemm_validate(ctypes.byref(p)) == 0
    # a shape NO table covers (24 x 48 x 48) records with the rule alone, and every launch of it is one the library would accept
    monkeypatch.setattr(ops, "_TUNED", {})
    e48 = UNetEngine(cfg, sd, 2, 24, 48, 48, 77, dev, n_t=1, share_prefix=True, packed=e64.packed)
    assert e48.n_tuned == 0 and e48.n_ruled >= 60, e48.n_ruled
    for op, p in e48.S.recorded:
        if op == L.OP_GEMM:
            assert e48.S.lib.vmv_gemm_validate(ctypes.byref(p)) == 0

"""`gpu`: the OpenCLIP text tower on the HIP kernels against the oracle restatement (oracle/clip_text.py; parity unpinned
against open_clip itself, which is absent) — a small config and the full ViT-H/14 text side (24 layers, width 1024, 16 heads,
354 M parameters, random-init: there are no weights offline), penultimate layer as VideoMV configures it."""
import pytest
import torch

from oracle.clip_text import text_tower
from tests.test_clip_cpu import SMALL, random_sd, tokens_for, rel_l2
from videomv_amd import _lib as L
from videomv_amd.clip_text import ClipTextEngine, ClipTextOptions

pytestmark = pytest.mark.gpu
TOL = 6e-3 if L.elem() == torch.float16 else 4e-2


def test_small_tower_matches_oracle():
    o, B = SMALL, 3
    sd = random_sd(o, 11)
    tok = tokens_for(o, B, 12)
    taps, taps_ref = {}, {}
    eng = ClipTextEngine(o, sd, B, torch.device("cuda"), layer_idx=1, taps=taps)
    xt, x = eng.forward(tok)
    xt_o, x_o = text_tower(sd, tok, o.width, o.heads, o.layers, layer_idx=1, taps=taps_ref)
    for k, a in taps.items():
        assert rel_l2(a.tensor().float().cpu().view(B, o.context_length, o.width), taps_ref[k]) < TOL, k
    assert rel_l2(x.cpu(), x_o) < TOL and rel_l2(xt.cpu(), xt_o) < TOL
    xt2, x2 = eng.forward(tok)                                      # replay: same plan, same answer
    assert torch.equal(x, x2) and torch.equal(xt, xt2)


def test_vit_h14_text_tower_matches_oracle():
    o, B = ClipTextOptions(), 2
    sd = random_sd(o, 21)
    tok = tokens_for(o, B, 22)
    eng = ClipTextEngine(o, sd, B, torch.device("cuda"), layer_idx=1)
    xt, x = eng.forward(tok)
    xt_o, x_o = text_tower(sd, tok, o.width, o.heads, o.layers, layer_idx=1)
    assert x.shape == (B, 77, 1024) and xt.shape == (B, 1024)
    assert torch.isfinite(x).all()
    assert rel_l2(x.cpu(), x_o) < 2 * TOL, rel_l2(x.cpu(), x_o)        # 23 blocks of 16-bit residual stream
    assert rel_l2(xt.cpu(), xt_o) < 2 * TOL, rel_l2(xt.cpu(), xt_o)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        eng.S.run()
    e1.record(); torch.cuda.synchronize()
    print(f"\nViT-H/14 text tower, {B} prompts: {e0.elapsed_time(e1) / 5:.3f} ms per forward ({eng.S.nops} launches)")


# ------------------------------------------------------------------------------------------------------- image tower
from oracle.clip_vision import image_tower
from tests.test_clip_cpu import VSMALL
from videomv_amd.clip_vision import ClipVisionEngine, ClipVisionOptions, clip_vision_shapes


def test_small_vision_tower_matches_oracle():
    o, B = VSMALL, 3
    sd = random_sd(o, 51, clip_vision_shapes(o))
    img = torch.randn(B, 3, o.image_size, o.image_size, generator=torch.Generator().manual_seed(52))
    taps, taps_ref = {}, {}
    eng = ClipVisionEngine(o, sd, B, torch.device("cuda"), taps=taps)
    out = eng.forward(img)
    ref = image_tower(sd, img, o.width, o.heads, o.layers, o.patch_size, taps=taps_ref)
    for k, a in taps.items():
        assert rel_l2(a.tensor().float().cpu().view(B, eng.T, o.width), taps_ref[k]) < TOL, k
    assert rel_l2(out.cpu(), ref) < TOL
    assert torch.equal(out, eng.forward(img))


def test_vit_h14_image_tower_matches_oracle():
    o, B = ClipVisionOptions(), 1
    sd = random_sd(o, 61, clip_vision_shapes(o))
    img = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(62))
    eng = ClipVisionEngine(o, sd, B, torch.device("cuda"))
    out = eng.forward(img)
    ref = image_tower(sd, img, o.width, o.heads, o.layers, o.patch_size)
    assert out.shape == (B, 1024) and torch.isfinite(out).all()
    assert rel_l2(out.cpu(), ref) < 3 * TOL, rel_l2(out.cpu(), ref)          # 32 blocks of 16-bit residual stream, one pooled token
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        eng.S.run()
    e1.record(); torch.cuda.synchronize()
    print(f"\nViT-H/14 image tower, {B} image: {e0.elapsed_time(e1) / 5:.3f} ms per forward ({eng.S.nops} launches)")

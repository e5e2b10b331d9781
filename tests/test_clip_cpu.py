"""`not gpu`: the OpenCLIP text tower.  (1) The oracle restatement against the same published block structure
assembled from torch.nn.MultiheadAttention / LayerNorm / GELU (the package itself is absent: parity unpinned against it).
(2) The recorded plan of ClipTextEngine, interpreted on CPU, against the oracle — this pins the launch arguments (fused q|k|v
layout, causal flag, GELU activation, residuals, the penultimate-layer cut)."""
import dataclasses

import pytest
import torch

from oracle.clip_text import text_tower
from tests import plan_interp
from videomv_amd import _lib as L
from videomv_amd.clip_text import ClipTextOptions, clip_text_shapes

SMALL = ClipTextOptions(vocab_size=300, context_length=77, width=128, heads=2, layers=4, embed_dim=96)


def random_sd(o, seed, shapes=None):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in (shapes or clip_text_shapes(o)).items():
        if ".ln_" in k and k.endswith(".weight") or k == "ln_final.weight":
            sd[k] = 1 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith("bias"):
            sd[k] = 0.05 * torch.randn(shp, generator=g)
        elif k in ("token_embedding.weight", "positional_embedding", "visual.positional_embedding", "visual.class_embedding"):
            sd[k] = 0.5 * torch.randn(shp, generator=g)
        else:
            sd[k] = torch.randn(shp, generator=g) * shp[-1] ** -0.5
    return sd


def tokens_for(o, B, seed):
    g = torch.Generator().manual_seed(seed)
    t = torch.zeros(B, o.context_length, dtype=torch.long)
    for b in range(B):
        n = int(torch.randint(3, o.context_length - 2, (1,), generator=g))
        t[b, 0] = o.vocab_size - 2                                   # <start_of_text>
        t[b, 1:n] = torch.randint(1, o.vocab_size - 2, (n - 1,), generator=g)
        t[b, n] = o.vocab_size - 1                                   # <end_of_text>: the largest id
    return t


class _Block(torch.nn.Module):
    """open_clip's ResidualAttentionBlock as published: ln_1, nn.MultiheadAttention, ln_2, mlp (c_fc, gelu, c_proj)."""

    def __init__(self, W, heads):
        super().__init__()
        self.ln_1, self.ln_2 = torch.nn.LayerNorm(W), torch.nn.LayerNorm(W)
        self.attn = torch.nn.MultiheadAttention(W, heads)
        self.mlp = torch.nn.Sequential()
        self.mlp.add_module("c_fc", torch.nn.Linear(W, 4 * W))
        self.mlp.add_module("gelu", torch.nn.GELU())
        self.mlp.add_module("c_proj", torch.nn.Linear(4 * W, W))

    def forward(self, x, attn_mask):          # x: [T, B, W] (LND, clip_embedder.py:195)
        h = self.ln_1(x)
        x = x + self.attn(h, h, h, need_weights=False, attn_mask=attn_mask)[0]
        return x + self.mlp(self.ln_2(x))


def test_oracle_matches_torch_modules():
    o = SMALL
    sd = random_sd(o, 5)
    tok = tokens_for(o, 3, 6)
    blocks = torch.nn.ModuleList([_Block(o.width, o.heads) for _ in range(o.layers)])
    blocks.load_state_dict({k[len("transformer.resblocks."):]: v for k, v in sd.items() if k.startswith("transformer.resblocks.")})
    ln_final = torch.nn.LayerNorm(o.width)
    ln_final.load_state_dict({"weight": sd["ln_final.weight"], "bias": sd["ln_final.bias"]})
    with torch.no_grad():
        x = (sd["token_embedding.weight"][tok] + sd["positional_embedding"]).permute(1, 0, 2)
        mask = torch.empty(o.context_length, o.context_length).fill_(float("-inf")).triu_(1)
        for i, r in enumerate(blocks):
            if i == len(blocks) - 1:                              # layer="penultimate" (clip_embedder.py:219-220)
                break
            x = r(x, mask)
        x = ln_final(x.permute(1, 0, 2))
        xt = x[torch.arange(3), tok.argmax(dim=-1)] @ sd["text_projection"]
    xt_o, x_o = text_tower(sd, tok, o.width, o.heads, o.layers, layer_idx=1)
    assert float((x - x_o).abs().max()) < 2e-5 and float((xt - xt_o).abs().max()) < 2e-5
    x_last = text_tower(sd, tok, o.width, o.heads, o.layers, layer_idx=0)[1]
    assert float((x_last - x_o).abs().max()) > 1e-2             # the cut matters


def rel_l2(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12))


@pytest.mark.parametrize("layer_idx", [1, 0])
def test_plan_matches_oracle(monkeypatch, layer_idx):
    plan_interp.install(monkeypatch)
    from videomv_amd.clip_text import ClipTextEngine
    o = SMALL
    sd = random_sd(o, 7)
    B = 2
    tok = tokens_for(o, B, 8)
    taps, taps_ref = {}, {}
    eng = ClipTextEngine(o, sd, B, torch.device("cpu"), layer_idx=layer_idx, taps=taps)
    xt, x = eng.forward(tok)
    xt_o, x_o = text_tower(sd, tok, o.width, o.heads, o.layers, layer_idx=layer_idx, taps=taps_ref)
    tol = 6e-3 if L.elem() == torch.float16 else 3e-2
    for k, a in taps.items():
        assert rel_l2(a.tensor().view(B, o.context_length, o.width), taps_ref[k]) < tol, k
    assert x.shape == x_o.shape and rel_l2(x, x_o) < tol, rel_l2(x, x_o)
    assert xt.shape == xt_o.shape and rel_l2(xt, xt_o) < tol, rel_l2(xt, xt_o)
    n_layers = o.layers - layer_idx
    assert eng.S.nops == 7 * n_layers + 1                        # 2 LN + 4 GEMM + 1 attention per block, ln_final
    with pytest.raises(ValueError):
        eng.forward(tok[:, :-1])
    bad = tok.clone(); bad[0, 3] = o.vocab_size
    with pytest.raises(ValueError):
        eng.forward(bad)


def test_full_size_shapes():
    s = clip_text_shapes(ClipTextOptions())
    assert s["transformer.resblocks.23.attn.in_proj_weight"] == (3072, 1024) and s["text_projection"] == (1024, 1024)
    import math
    assert sum(math.prod(v) for v in s.values()) == 354_032_640


def test_embedder_uses_the_tower_when_a_checkpoint_is_given(monkeypatch, tmp_path):
    """The reference-named embedder: synthetic stand-in without weights; with an open_clip-style checkpoint it runs the tower
    (token ids from a tokenizer callable), and says so when it has no tokenizer."""
    plan_interp.install(monkeypatch)
    from videomv_amd.embedder import FrozenOpenCLIPTtxtVisualEmbedder
    o = SMALL
    sd = random_sd(o, 9)
    sd["visual.proj"] = torch.zeros(4, 4)                                      # (an image side without conv1: ignored)
    path = tmp_path / "open_clip_pytorch_model.bin"
    torch.save(sd, path)
    tok = tokens_for(o, 1, 10)
    syn = FrozenOpenCLIPTtxtVisualEmbedder(pretrained="/nonexistent/open_clip_pytorch_model.bin")
    assert syn(text=["a chair"])[2].shape == (1, 77, 1024)
    emb = FrozenOpenCLIPTtxtVisualEmbedder(pretrained=str(path), layer="penultimate", tokenizer=lambda texts: tok, device="cpu")
    _, xt, x = emb(text=["a chair"])
    xt_o, x_o = text_tower(sd, tok, o.width, o.heads, o.layers, layer_idx=1)
    assert rel_l2(x, x_o) < 3e-2 and rel_l2(xt, xt_o) < 3e-2
    none = FrozenOpenCLIPTtxtVisualEmbedder(pretrained=str(path), device="cpu")
    none.tokenizer = None
    with pytest.raises(RuntimeError, match="tokenizer"):
        none(text=["a chair"])
    assert torch.equal(none(tokens=tok)[2], x)


# ------------------------------------------------------------------------------------------------------- image tower
from oracle.clip_vision import image_tower
from videomv_amd.clip_vision import ClipVisionOptions, clip_vision_shapes

VSMALL = ClipVisionOptions(image_size=56, patch_size=14, width=160, heads=2, layers=3, mlp_ratio=2.0, embed_dim=48)      # head_dim 80 like ViT-H/14


def test_vision_oracle_matches_torch_modules():
    o = VSMALL
    sd = random_sd(o, 31, clip_vision_shapes(o))
    img = torch.randn(2, 3, o.image_size, o.image_size, generator=torch.Generator().manual_seed(32))
    blocks = torch.nn.ModuleList([_Block(o.width, o.heads) for _ in range(o.layers)])
    pre = "visual.transformer.resblocks."
    bsd = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    hid = int(o.width * o.mlp_ratio)
    for b in blocks:                                              # (_Block builds a 4x MLP; this config uses mlp_ratio 2)
        b.mlp.c_fc, b.mlp.c_proj = torch.nn.Linear(o.width, hid), torch.nn.Linear(hid, o.width)
    blocks.load_state_dict(bsd)
    with torch.no_grad():
        x = torch.nn.functional.conv2d(img, sd["visual.conv1.weight"], stride=o.patch_size).reshape(2, o.width, -1).permute(0, 2, 1)
        x = torch.cat([sd["visual.class_embedding"].expand(2, 1, o.width), x], dim=1) + sd["visual.positional_embedding"]
        x = torch.nn.functional.layer_norm(x, (o.width,), sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])
        x = x.permute(1, 0, 2)
        for r in blocks:
            x = r(x, None)
        x = x.permute(1, 0, 2)
        ref = torch.nn.functional.layer_norm(x[:, 0], (o.width,), sd["visual.ln_post.weight"], sd["visual.ln_post.bias"]) @ sd["visual.proj"]
    got = image_tower(sd, img, o.width, o.heads, o.layers, o.patch_size)
    assert float((ref - got).abs().max()) < 2e-5


def test_vision_plan_matches_oracle(monkeypatch):
    """head_dim 80 packed as 128 (zero rows / columns), patch GEMM with K padded 588 -> 592, class token, ln_pre / ln_post."""
    plan_interp.install(monkeypatch)
    from videomv_amd.clip_vision import ClipVisionEngine
    o = VSMALL
    sd = random_sd(o, 33, clip_vision_shapes(o))
    B = 2
    img = torch.randn(B, 3, o.image_size, o.image_size, generator=torch.Generator().manual_seed(34))
    taps, taps_ref = {}, {}
    eng = ClipVisionEngine(o, sd, B, torch.device("cpu"), taps=taps)
    assert eng.hd == 80 and eng.hdp == 128 and eng.T == 17
    out = eng.forward(img)
    ref = image_tower(sd, img, o.width, o.heads, o.layers, o.patch_size, taps=taps_ref)
    tol = 6e-3 if L.elem() == torch.float16 else 3e-2
    for k, a in taps.items():
        assert rel_l2(a.tensor().view(B, eng.T, o.width), taps_ref[k]) < tol, k
    assert out.shape == ref.shape and rel_l2(out, ref) < tol, rel_l2(out, ref)
    with pytest.raises(ValueError):
        eng.forward(img[:, :, :-14])


def test_vision_full_size_shapes():
    import math
    s = clip_vision_shapes(ClipVisionOptions())
    assert s["visual.positional_embedding"] == (257, 1280) and s["visual.proj"] == (1280, 1024)
    assert sum(math.prod(v) for v in s.values()) == 632_076_800


def test_embedder_runs_both_towers(monkeypatch, tmp_path):
    plan_interp.install(monkeypatch)
    from videomv_amd.embedder import FrozenOpenCLIPTtxtVisualEmbedder
    ot, ov = SMALL, dataclasses.replace(VSMALL, width=128, heads=2, embed_dim=96)      # width 128 -> the embedder's head_dim-64 rule
    sd = random_sd(ot, 41)
    sd.update(random_sd(ov, 42, clip_vision_shapes(ov)))
    path = tmp_path / "open_clip_pytorch_model.bin"
    torch.save({"state_dict": sd}, path)
    tok = tokens_for(ot, 1, 43)
    img = torch.randn(1, 3, ov.image_size, ov.image_size, generator=torch.Generator().manual_seed(44))
    emb = FrozenOpenCLIPTtxtVisualEmbedder(pretrained=str(path), tokenizer=lambda t: tok, device="cpu")
    y_visual, xt, x = emb(text=[""], image=img)
    ref = image_tower(sd, img, ov.width, ov.heads, ov.layers, ov.patch_size)
    assert y_visual.shape == (1, 96) and rel_l2(y_visual, ref) < 3e-2
    assert x.shape == (1, 77, 128)


# ------------------------------------------------------------------------------------------------------- tokenizer
def test_bpe_tokenizer_on_a_synthetic_merges_file(tmp_path):
    """The published CLIP BPE on a small merges file: vocabulary layout, merge priority, '</w>' word ends, lower-casing and
    whitespace collapse, apostrophe / digit / punctuation splitting, byte-level fallback for non-ASCII, framing, truncation."""
    import gzip
    from videomv_amd.clip_tokenizer import ClipBpeTokenizer, byte_symbols
    merges = ["c h", "ch a", "i r</w>", "cha ir</w>", "a n", "t h", "th e</w>", "an d</w>", "Ã ©</w>"]
    path = tmp_path / "bpe.txt.gz"
    with gzip.open(path, "wb") as f:
        f.write(("#version: synthetic\n" + "\n".join(merges) + "\n").encode("utf-8"))
    tok = ClipBpeTokenizer(str(path), context_length=12, vocab_size=512 + len(merges) + 2)
    assert len(tok.ids) == 512 + len(merges) + 2 and tok.sot == 512 + len(merges) and tok.eot == tok.sot + 1
    sym = byte_symbols()
    assert len(set(sym.values())) == 256 and sym[ord("a")] == "a" and sym[ord(" ")] == chr(256 + 32)      # space is remapped
    assert tok.encode("Chair") == [tok.ids["chair</w>"]]                                   # c h -> ch a -> (i r</w>) -> cha ir</w>
    assert tok.encode("  the   CHAIR and\tthe chairs ") == [tok.ids["the</w>"], tok.ids["chair</w>"], tok.ids["and</w>"], tok.ids["the</w>"],
                                                            tok.ids["cha"], tok.ids["i"], tok.ids["r"], tok.ids["s</w>"]]
    assert tok.encode("it's 42!") == [tok.ids["i"], tok.ids["t</w>"], tok.ids["'"], tok.ids["s</w>"], tok.ids["4</w>"], tok.ids["2</w>"], tok.ids["!</w>"]]
    assert tok.encode("\u00e9") == [tok.ids["\u00c3\u00a9</w>"]]                            # UTF-8 bytes C3 A9, merged by the last rule
    assert tok.encode("a &amp;amp; b") == [tok.ids["a</w>"], tok.ids["&</w>"], tok.ids["b</w>"]]     # html.unescape twice
    t = tok(["the chair", "and " * 20])
    assert t.shape == (2, 12) and t.dtype == torch.long
    assert t[0].tolist() == [tok.sot, tok.ids["the</w>"], tok.ids["chair</w>"], tok.eot] + [0] * 8
    assert t[1, 0] == tok.sot and t[1, -1] == tok.eot and (t[1, 1:-1] == tok.ids["and</w>"]).all()      # cut, <end> kept last
    assert t.argmax(dim=-1).tolist() == [3, 11]                                              # what the pooled feature indexes

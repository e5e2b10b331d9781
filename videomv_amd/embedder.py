"""Text-conditioning inputs.  The sampler needs ``y_words`` [B, 77, 1024] (and, for I2VGen, ``y_visual`` [B, 1024]) — the outputs
of ``FrozenOpenCLIPTtxtVisualEmbedder`` (tools/modules/clip_embedder.py:144-227), which runs once per prompt, upstream of the hot
path.

* TEXT tower: ``clip_text.ClipTextEngine`` runs open_clip's ViT-H/14 text transformer on the HIP kernels when ``pretrained``
  names an open_clip checkpoint (a state dict holding ``token_embedding.weight`` ...).  Tokenisation needs open_clip's BPE
  vocabulary, which is not available offline: token ids come from ``clip_tokenizer.ClipBpeTokenizer`` when ``bpe_path`` names the
  merges file, from ``open_clip.tokenize`` when that package is importable, from a ``tokenizer=`` callable, or are passed directly
  (``forward(tokens=...)``).
* IMAGE tower: ``clip_vision.ClipVisionEngine`` (ViT-H/14 visual, head_dim 80 packed 128 wide for the flash kernel) when the
  checkpoint holds the ``visual.*`` keys: ``y_visual = encode_image(image)`` for I2VGen (inference_i2vgen_entrance.py:246).
* Without a checkpoint (this container: no weights, no network) ``SyntheticTextEmbedder`` produces deterministic, prompt-seeded
  stand-ins of the right shapes so the drop-in entrance, the benchmark and CI run end-to-end; recorded CLIP features can be fed
  through ``features_path`` (a .pt file mapping prompt -> tensor)."""
import hashlib
import os

import torch

from .registry import EMBEDDER


class _TextFeatures(torch.nn.Module):
    def __init__(self, tokens=77, width=1024, features_path=None, **kwargs):
        super().__init__()
        self.tokens, self.width = tokens, width
        self.table = torch.load(features_path, map_location="cpu") if features_path else None

    def forward(self, text=None, image=None):
        """-> (y_visual [B,W] or None, y_text [B,1,W], y_words [B,T,W]) like FrozenOpenCLIPTtxtVisualEmbedder (:145-227).
        ``image`` (a normalised image tensor) yields a deterministic stand-in for the CLIP image embedding."""
        if isinstance(text, str):
            text = [text]
        y_visual = None
        if image is not None:
            vis = []
            for im in image:
                seed = int.from_bytes(hashlib.sha256(im.detach().cpu().float().numpy().tobytes()).digest()[:8], "little") % (2 ** 63)
                vis.append(torch.randn(self.width, generator=torch.Generator().manual_seed(seed)))
            y_visual = torch.stack(vis, 0)
        outs = []
        for t in text:
            if self.table is not None and t in self.table:
                outs.append(self.table[t].float().reshape(self.tokens, self.width))
                continue
            seed = int.from_bytes(hashlib.sha256(t.encode("utf-8")).digest()[:8], "little") % (2 ** 63)
            g = torch.Generator().manual_seed(seed)
            outs.append(torch.randn(self.tokens, self.width, generator=g))
        y_words = torch.stack(outs, 0)
        return y_visual, y_words.mean(dim=1, keepdim=True), y_words


@EMBEDDER.register_class()
class SyntheticTextEmbedder(_TextFeatures):
    pass


@EMBEDDER.register_class()
class FrozenOpenCLIPTtxtVisualEmbedder(_TextFeatures):
    """Registered under the reference's name so unchanged YAMLs build.  ``pretrained`` = an open_clip checkpoint -> the real text
    tower on the HIP kernels (module docstring); otherwise synthetic / recorded features."""

    def __init__(self, pretrained=None, layer="penultimate", vit_resolution=(224, 224), tokenizer=None, device="cuda", bpe_path=None, **kwargs):
        super().__init__(features_path=kwargs.pop("features_path", None))
        if layer not in ("last", "penultimate"):
            raise NotImplementedError(layer)                                   # (clip_embedder.py:165-170)
        self.layer_idx = 1 if layer == "penultimate" else 0
        self.tokenizer, self.tower_device = tokenizer, device
        self.tower_sd, self.visual_sd, self._towers = None, None, {}
        if pretrained and os.path.isfile(str(pretrained)):
            sd = torch.load(pretrained, map_location="cpu")
            sd = sd.get("state_dict", sd)
            if "token_embedding.weight" in sd:
                self.tower_sd = {k: v for k, v in sd.items() if not k.startswith("visual.")}
            if "visual.conv1.weight" in sd and "visual.proj" in sd:
                self.visual_sd = {k: v for k, v in sd.items() if k.startswith("visual.")}
        if self.tokenizer is None and bpe_path:                                # open_clip's merges file, wherever the deployment keeps it
            from .clip_tokenizer import ClipBpeTokenizer
            self.tokenizer = ClipBpeTokenizer(bpe_path)
        if self.tokenizer is None:
            try:
                import open_clip                                               # absent in this image
                self.tokenizer = open_clip.tokenize
            except ImportError:
                pass

    def _tower(self, B):
        from .clip_text import ClipTextEngine, ClipTextOptions
        if B not in self._towers:
            sd = self.tower_sd
            o = ClipTextOptions(vocab_size=sd["token_embedding.weight"].shape[0], context_length=sd["positional_embedding"].shape[0],
                                width=sd["positional_embedding"].shape[1], heads=sd["positional_embedding"].shape[1] // 64,
                                layers=1 + max(int(k.split(".")[2]) for k in sd if k.startswith("transformer.resblocks.")),
                                embed_dim=sd["text_projection"].shape[1])
            donor = next((e for k, e in self._towers.items() if not isinstance(k, tuple)), None)      # packed weights: one copy for all batch sizes
            self._towers[B] = ClipTextEngine(o, sd, B, torch.device(self.tower_device), layer_idx=self.layer_idx, donor=donor)
        return self._towers[B]

    def _vision(self, B):
        from .clip_vision import ClipVisionEngine, ClipVisionOptions
        if ("v", B) not in self._towers:
            sd = self.visual_sd
            W = sd["visual.conv1.weight"].shape[0]
            ps = sd["visual.conv1.weight"].shape[-1]
            g = int(round((sd["visual.positional_embedding"].shape[0] - 1) ** 0.5))
            hd = 80 if W == 1280 else 64                                       # (ViT-H/14: 16 heads of 80; the B / L families use 64)
            o = ClipVisionOptions(image_size=g * ps, patch_size=ps, width=W, heads=W // hd,
                                  layers=1 + max(int(k.split(".")[3]) for k in sd if k.startswith("visual.transformer.resblocks.")),
                                  mlp_ratio=sd["visual.transformer.resblocks.0.mlp.c_fc.weight"].shape[0] / W, embed_dim=sd["visual.proj"].shape[1])
            donor = next((e for k, e in self._towers.items() if isinstance(k, tuple)), None)
            self._towers[("v", B)] = ClipVisionEngine(o, sd, B, torch.device(self.tower_device), donor=donor)
        return self._towers[("v", B)]

    def forward(self, text=None, image=None, tokens=None):
        if self.tower_sd is None:
            return super().forward(text=text, image=image)
        if tokens is None:
            if self.tokenizer is None:
                raise RuntimeError("CLIP weights are loaded but there is no tokenizer (open_clip's BPE vocabulary is not bundled): "
                                   "pass tokens=open_clip.tokenize(text) or construct the embedder with tokenizer=")
            tokens = self.tokenizer([text] if isinstance(text, str) else list(text))
        xt, x = self._tower(tokens.shape[0]).forward(tokens)
        y_visual = None
        if image is not None:                                                  # (clip_embedder.py:187)
            y_visual = self._vision(image.shape[0]).forward(image) if self.visual_sd is not None else \
                super().forward(text=[""] * tokens.shape[0], image=image)[0]
        return y_visual, xt, x


@EMBEDDER.register_class()
class FrozenOpenCLIPEmbedder(FrozenOpenCLIPTtxtVisualEmbedder):
    pass

"""Text-conditioning inputs.  The OpenCLIP ViT-H/14 text tower (tools/modules/clip_embedder.py) is OUT OF SCOPE for the
hot path (it runs once per prompt) and its package/weights are not available offline; the sampler only needs
its output ``y_words`` [1, 77, 1024].  ``SyntheticTextEmbedder`` produces a deterministic, prompt-seeded stand-in of that
shape so the drop-in entrance, the benchmark and CI can run end-to-end; recorded CLIP features can be fed instead via
``cfg.text_features`` (a .pt file mapping prompt -> tensor)."""
import hashlib

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
    """Registered under the reference's name so unchanged YAMLs build; produces synthetic / recorded features (see the
    module docstring) — the real CLIP towers are not part of this implementation."""

    def __init__(self, pretrained=None, layer="penultimate", vit_resolution=(224, 224), **kwargs):
        super().__init__(features_path=kwargs.pop("features_path", None))


@EMBEDDER.register_class()
class FrozenOpenCLIPEmbedder(FrozenOpenCLIPTtxtVisualEmbedder):
    pass

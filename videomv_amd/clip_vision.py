"""OpenCLIP ViT-H/14 IMAGE tower on the gfx950 kernels — the I2VGen conditioning ``y_visual``.

Reference: ``FrozenOpenCLIPTtxtVisualEmbedder.forward`` calls ``self.model.encode_image(image)`` (tools/modules/clip_embedder.py:187);
the module behind it is ``open_clip``'s ``VisionTransformer`` (absent here; restated from its published structure in
``oracle/clip_vision.py``, parity unpinned against the package): 14 x 14 patch convolution (no bias), class token, positional
embedding, ``ln_pre``, 32 ``ResidualAttentionBlock``s (width 1280, 16 heads of 80, MLP 5120, no mask), ``ln_post`` of the class
token, ``@ proj`` -> [B, 1024].

The flash kernel has head_dim 32 / 64 / 128: the 80-wide heads are packed 128 wide with zero rows in the q | k | v projection and
zero columns in out_proj (``_ClipTower._pack_block``), which changes neither the scores nor the outputs.  The patch gather
(``unfold``), the class-token / positional-embedding assembly and the class-token gather are data moves in PyTorch; every
FLOP of the transformer runs in the recorded plan.
"""
import dataclasses
from typing import Dict, Optional

import torch

from . import _lib as L
from . import ops
from . import packing as P
from .unet_engine import Act
from .clip_text import _ClipTower


@dataclasses.dataclass
class ClipVisionOptions:
    """Vision side of open_clip's ``ViT-H-14`` model config."""
    image_size: int = 224
    patch_size: int = 14
    width: int = 1280
    heads: int = 16
    layers: int = 32
    mlp_ratio: float = 4.0
    embed_dim: int = 1024


def clip_vision_shapes(o: ClipVisionOptions) -> Dict[str, tuple]:
    """State-dict names / shapes of ``open_clip.CLIP.visual`` (keys as in ``model.state_dict()``, i.e. with the ``visual.`` prefix)."""
    W, g = o.width, o.image_size // o.patch_size
    H = int(W * o.mlp_ratio)
    s = {"visual.conv1.weight": (W, 3, o.patch_size, o.patch_size), "visual.class_embedding": (W,),
         "visual.positional_embedding": (g * g + 1, W), "visual.ln_pre.weight": (W,), "visual.ln_pre.bias": (W,),
         "visual.ln_post.weight": (W,), "visual.ln_post.bias": (W,), "visual.proj": (W, o.embed_dim)}
    for i in range(o.layers):
        p = f"visual.transformer.resblocks.{i}."
        s.update({p + "ln_1.weight": (W,), p + "ln_1.bias": (W,), p + "attn.in_proj_weight": (3 * W, W),
                  p + "attn.in_proj_bias": (3 * W,), p + "attn.out_proj.weight": (W, W), p + "attn.out_proj.bias": (W,),
                  p + "ln_2.weight": (W,), p + "ln_2.bias": (W,), p + "mlp.c_fc.weight": (H, W), p + "mlp.c_fc.bias": (H,),
                  p + "mlp.c_proj.weight": (W, H), p + "mlp.c_proj.bias": (W,)})
    return s


def _padded_head_dim(hd: int) -> int:
    for d in (32, 64, 128):
        if hd <= d:
            return d
    raise NotImplementedError(f"head_dim {hd} > 128")


class ClipVisionEngine(_ClipTower):
    """Plan for ``B`` images: normalised pixels [B, 3, S, S] -> image embedding fp32 [B, embed_dim]."""

    _PACKED_ATTRS = ("kpatch", "cls", "pos")

    def __init__(self, opt: ClipVisionOptions, sd: Dict[str, torch.Tensor], B: int, device, taps: Optional[dict] = None, donor=None):
        if opt.width % opt.heads or (opt.width // opt.heads) % 8 or opt.image_size % opt.patch_size:
            raise ValueError("width / heads must be a multiple of 8 and the image a whole number of patches")
        self.o, self.B = opt, int(B)
        self.hd = opt.width // opt.heads
        self.hdp = _padded_head_dim(self.hd)
        self.grid = opt.image_size // opt.patch_size
        self.T = self.grid * self.grid + 1
        self._init_common(device, taps)
        self._pack_or_share(sd, donor)
        self._build()

    def _pack(self, sd):
        dev, w, o = self.device, self.wt, self.o
        v = lambda k: sd["visual." + k]
        w["patch.weight"] = P.pack_linear(v("conv1.weight").detach().float().reshape(o.width, -1), dev)     # [W][3 * p * p], K padded to 8
        self.kpatch = w["patch.weight"].shape[1]
        self.cls = v("class_embedding").detach().float().to(dev)
        self.pos = v("positional_embedding").detach().float().to(dev)
        for n in ("ln_pre", "ln_post"):
            w[n + ".weight"], w[n + ".bias"] = P.f32(v(n + ".weight"), dev), P.f32(v(n + ".bias"), dev)
        for i in range(o.layers):
            p = f"transformer.resblocks.{i}."
            self._pack_block(sd, "visual." + p, p, o.heads, self.hd, self.hdp)
        w["proj.weight"] = P.pack_linear(v("proj").detach().float().t().contiguous(), dev)

    def _build(self):
        o, B, T, W = self.o, self.B, self.T, self.o.width
        dev = self.device
        np_ = self.grid * self.grid
        self.patches = torch.zeros(B * np_, self.kpatch, dtype=L.elem(), device=dev)       # im2col rows (cols >= 3 p p stay zero)
        self.tok = torch.zeros(B * np_, W, dtype=torch.float32, device=dev)                 # patch embeddings
        self.E = ops.Stream(record=False)
        self.patch_params = ops.gemm_params(B * np_, W, ops.linear_segs([(self.patches.data_ptr(), self.kpatch, self.kpatch)]),
                                            self.wt["patch.weight"], self.tok.data_ptr(), W, out_fp32=True)
        self.x_rows = torch.zeros(B * T, W, dtype=L.elem(), device=dev)
        x0 = Act(self.x_rows.view(torch.uint8).view(-1), B * T, W)
        x = self._ln("ln_pre", x0, "ln_pre")
        for i in range(o.layers):
            x = self._block(f"transformer.resblocks.{i}.", x, B, T, o.heads, self.hd, self.hdp, False)
            if self.taps is not None:
                self.taps[f"resblocks.{i}"] = x
        self.out = x

    @torch.no_grad()
    def forward(self, image: torch.Tensor) -> torch.Tensor:
        o, B, T, W = self.o, self.B, self.T, self.o.width
        if tuple(image.shape) != (B, 3, o.image_size, o.image_size):
            raise ValueError(f"image must be [{B}, 3, {o.image_size}, {o.image_size}], got {tuple(image.shape)}")
        ps = o.patch_size
        cols = torch.nn.functional.unfold(image.to(self.device).float(), kernel_size=ps, stride=ps)      # [B, 3 p p, grid^2], (c, ky, kx) major
        self.patches[:, : 3 * ps * ps].copy_(cols.transpose(1, 2).reshape(-1, 3 * ps * ps))
        self.E.gemm(self.patch_params, "patch_embed")
        xr = self.x_rows.view(B, T, W)
        xr[:, 0] = (self.cls + self.pos[0]).to(L.elem())
        xr[:, 1:] = (self.tok.view(B, T - 1, W) + self.pos[1:]).to(L.elem())
        self.S.run()
        cls_rows = self.out.tensor().view(B, T, W)[:, 0].contiguous()                      # class token (pool_type 'tok')
        pooled = torch.empty(B, W, dtype=L.elem(), device=self.device)
        self.E.layernorm(ops.ln_params(cls_rows.data_ptr(), W, pooled.data_ptr(), W, self.wt["ln_post.weight"], self.wt["ln_post.bias"], B, W, 1e-5), "ln_post")
        out = torch.empty(B, o.embed_dim, dtype=torch.float32, device=self.device)
        self.E.gemm(ops.gemm_params(B, self.wt["proj.weight"].shape[0], ops.linear_segs([(pooled.data_ptr(), W, W)]), self.wt["proj.weight"],
                                    out.data_ptr(), o.embed_dim, out_fp32=True), "proj")
        return out

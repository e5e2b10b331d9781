"""OpenCLIP ViT-H/14 TEXT tower on the gfx950 kernels — the once-per-prompt caller upstream of the hot path.

Reference: ``FrozenOpenCLIPTtxtVisualEmbedder.encode_with_transformer`` / ``text_transformer_forward``
(``tools/modules/clip_embedder.py:192-201, 217-225``; ``FrozenOpenCLIPEmbedder`` :113-138 is the same walk without the pooled
vector): token embedding + positional embedding, the transformer's residual attention blocks except the last ``layer_idx``
(``layer="penultimate"`` -> 1), ``ln_final``, and ``xt = x[argmax token] @ text_projection``.  The blocks themselves are
``open_clip``'s ``ResidualAttentionBlock`` (pre-LN, ``nn.MultiheadAttention`` with the causal additive mask, MLP 4x with exact
GELU) — a pip dependency absent from /root/reference and from this image, restated from its published structure in
``oracle/clip_text.py`` (parity unpinned against the package itself; pinned against ``torch.nn.MultiheadAttention``).

``ClipTextEngine`` records the forward as a plan of C-ABI launches over 16-bit rows ``[B * 77, width]`` like the other engines:
LayerNorm -> fused q|k|v GEMM (+bias) -> causal flash attention (``VmvAttnParams.causal``) -> out-proj GEMM (+bias, +residual)
-> LayerNorm -> fc GEMM (+bias, ``VMV_ACT_GELU``) -> proj GEMM (+bias, +residual).  The embedding gather (a data move) and the
77-token argmax are the only PyTorch ops.  Tokenisation is NOT here: open_clip's BPE vocabulary file is not available offline,
so the tower takes token ids ``[B, 77]`` (``open_clip.tokenize`` output) — see ``embedder.py``.
"""
import dataclasses
from typing import Dict, Optional

import torch

from . import _lib as L
from . import ops
from . import packing as P
from .unet_engine import Pool, Act


@dataclasses.dataclass
class ClipTextOptions:
    """Text side of open_clip's ``ViT-H-14`` model config."""
    vocab_size: int = 49408
    context_length: int = 77
    width: int = 1024
    heads: int = 16
    layers: int = 24
    embed_dim: int = 1024


def clip_text_shapes(o: ClipTextOptions) -> Dict[str, tuple]:
    """State-dict names / shapes of the text side of ``open_clip.CLIP`` (the keys a ``model.state_dict()`` holds)."""
    W = o.width
    s = {"token_embedding.weight": (o.vocab_size, W), "positional_embedding": (o.context_length, W),
         "ln_final.weight": (W,), "ln_final.bias": (W,), "text_projection": (W, o.embed_dim)}
    for i in range(o.layers):
        p = f"transformer.resblocks.{i}."
        s.update({p + "ln_1.weight": (W,), p + "ln_1.bias": (W,), p + "attn.in_proj_weight": (3 * W, W),
                  p + "attn.in_proj_bias": (3 * W,), p + "attn.out_proj.weight": (W, W), p + "attn.out_proj.bias": (W,),
                  p + "ln_2.weight": (W,), p + "ln_2.bias": (W,), p + "mlp.c_fc.weight": (4 * W, W), p + "mlp.c_fc.bias": (4 * W,),
                  p + "mlp.c_proj.weight": (W, 4 * W), p + "mlp.c_proj.bias": (W,)})
    return s


class _ClipTower:
    """Shared by the text and image towers: pooled activations, the recorded stream, and open_clip's ResidualAttentionBlock
    (``x = x + attn(ln_1(x))``, ``x = x + c_proj(gelu(c_fc(ln_2(x))))``) as 7 launches."""

    def _init_common(self, device, taps):
        self.device, self.taps = device, taps
        self.pool = Pool(device)
        self.S = ops.Stream(record=True)
        self._splitk = ops.SplitK(device, cap=8)
        self.wt: Dict[str, torch.Tensor] = {}

    def _pack_or_share(self, sd, donor):
        """Packed weights are immutable and batch-size independent: an engine for another B of the same tower takes its donor's copy
        (one ~0.7 / 1.3 GB repack per batch size otherwise)."""
        if donor is not None and type(donor) is type(self) and str(donor.device) == str(self.device) and \
                getattr(donor, "layer_idx", None) == getattr(self, "layer_idx", None):
            self.wt = donor.wt
            for a in self._PACKED_ATTRS:
                setattr(self, a, getattr(donor, a))
        else:
            self._pack(sd)

    def act(self, rows, C, dtype=None) -> Act:
        return Act(self.pool.get(rows * C * (4 if dtype == torch.float32 else 2)), rows, C, dtype)

    def rel(self, a: Act):
        if self.taps is None and not getattr(a, "_pinned", False):
            self.pool.put(a.buf)

    def _gemm(self, label, x: Act, wkey, out: Act, **kw):
        Wt = self.wt[wkey + ".weight"]
        segs = ops.linear_segs([(x.ptr, x.C, x.C)])
        ks, ws = self._splitk.pick(x.rows, Wt.shape[0], segs)
        self.S.gemm(ops.gemm_params(x.rows, Wt.shape[0], segs, Wt, out.ptr, out.C, bias=self.wt.get(wkey + ".bias"),
                                    ksplit=ks, workspace=ws, **kw), label)

    def _ln(self, label, x: Act, key) -> Act:
        y = self.act(x.rows, x.C)
        self.S.layernorm(ops.ln_params(x.ptr, x.C, y.ptr, x.C, self.wt[key + ".weight"], self.wt[key + ".bias"], x.rows, x.C, 1e-5), label)
        return y

    def _pack_block(self, sd, src, p, heads, hd, hdp):
        """src: state-dict prefix of the block, p: packed-key prefix.  hdp > hd: every head's q / k / v rows (and the matching
        out_proj columns) are laid out hdp wide with zeros behind the hd real ones — scores and outputs are unchanged and the
        flash kernel sees a head_dim it has (80 -> 128)."""
        dev, w = self.device, self.wt
        for n in ("ln_1", "ln_2"):
            w[p + n + ".weight"], w[p + n + ".bias"] = P.f32(sd[src + n + ".weight"], dev), P.f32(sd[src + n + ".bias"], dev)
        W = heads * hd
        wi, bi = sd[src + "attn.in_proj_weight"].detach().float(), sd[src + "attn.in_proj_bias"].detach().float()
        wo = sd[src + "attn.out_proj.weight"].detach().float()
        if hdp != hd:
            wi_p = wi.new_zeros(3, heads, hdp, W); wi_p[:, :, :hd] = wi.view(3, heads, hd, W)
            bi_p = bi.new_zeros(3, heads, hdp); bi_p[:, :, :hd] = bi.view(3, heads, hd)
            wo_p = wo.new_zeros(W, heads, hdp); wo_p[:, :, :hd] = wo.view(W, heads, hd)
            wi, bi, wo = wi_p.view(3 * heads * hdp, W), bi_p.view(-1), wo_p.view(W, heads * hdp)
        w[p + "qkv.weight"], w[p + "qkv.bias"] = P.pack_linear(wi, dev), P.pack_bias(bi, dev)
        w[p + "out.weight"], w[p + "out.bias"] = P.pack_linear(wo, dev), P.pack_bias(sd[src + "attn.out_proj.bias"], dev)
        w[p + "fc.weight"], w[p + "fc.bias"] = P.pack_linear(sd[src + "mlp.c_fc.weight"], dev), P.pack_bias(sd[src + "mlp.c_fc.bias"], dev)
        w[p + "proj.weight"], w[p + "proj.bias"] = P.pack_linear(sd[src + "mlp.c_proj.weight"], dev), P.pack_bias(sd[src + "mlp.c_proj.bias"], dev)

    def _block(self, p, x: Act, B, T, heads, hd, hdp, causal, release_x=True) -> Act:
        rows, W, A = x.rows, x.C, heads * hdp
        h = self._ln(p + "ln_1", x, p + "ln_1")
        qkv = self.act(rows, 3 * A)
        self._gemm(p + "qkv", h, p + "qkv", qkv)
        self.rel(h)
        ao = self.act(rows, A)
        m = lambda: ops.seq_map(T * 3 * A, 0, 3 * A, inner=1)
        self.S.attention(ops.attn_params(qkv.ptr, qkv.ptr + 2 * A, qkv.ptr + 4 * A, ao.ptr, m(), m(), m(), ops.seq_map(T * A, 0, A, inner=1),
                                         B, heads, T, T, hd ** -0.5, head_dim=hdp, causal=causal), p + "attn")
        self.rel(qkv)
        y = self.act(rows, W)
        self._gemm(p + "out", ao, p + "out", y, residual=x.ptr, ldr=W)
        self.rel(ao)
        if release_x:
            self.rel(x)
        x = y
        h = self._ln(p + "ln_2", x, p + "ln_2")
        f = self.act(rows, self.wt[p + "fc.weight"].shape[0])
        self._gemm(p + "fc", h, p + "fc", f, act=L.ACT_GELU)
        self.rel(h)
        y = self.act(rows, W)
        self._gemm(p + "proj", f, p + "proj", y, residual=x.ptr, ldr=W)
        self.rel(f)
        self.rel(x)
        return y


class ClipTextEngine(_ClipTower):
    """Plan for ``B`` prompts: token ids [B, T] -> (xt fp32 [B, embed_dim], x fp32 [B, T, width])."""

    _PACKED_ATTRS = ("tok", "pos")          # what _pack leaves on the engine besides .wt (shared with a donor engine of another batch size)

    def __init__(self, opt: ClipTextOptions, sd: Dict[str, torch.Tensor], B: int, device, layer_idx: int = 1, taps: Optional[dict] = None,
                 donor=None):
        if opt.width % opt.heads or opt.width // opt.heads != 64:
            raise NotImplementedError("the flash kernel's causal path is head_dim 64 (ViT-H/14 text: 1024 / 16)")
        if not 0 <= layer_idx < opt.layers:
            raise ValueError("layer_idx")
        self.o, self.B, self.layer_idx = opt, int(B), int(layer_idx)
        self._init_common(device, taps)
        self._pack_or_share(sd, donor)
        self._build()

    # ------------------------------------------------------------------ weights
    def _pack(self, sd):
        dev, w, o = self.device, self.wt, self.o
        self.tok = sd["token_embedding.weight"].detach().float().to(dev)
        self.pos = sd["positional_embedding"].detach().float().to(dev)
        for i in range(o.layers - self.layer_idx):
            p = f"transformer.resblocks.{i}."
            self._pack_block(sd, p, p, o.heads, 64, 64)
        w["ln_final.weight"], w["ln_final.bias"] = P.f32(sd["ln_final.weight"], dev), P.f32(sd["ln_final.bias"], dev)
        w["text_projection"] = P.pack_linear(sd["text_projection"].detach().float().t().contiguous(), dev)      # x @ P = x . (P^T)^T

    # ------------------------------------------------------------------ plan
    def _build(self):
        o, B = self.o, self.B
        T, W = o.context_length, o.width
        rows = B * T
        self.x_rows = torch.zeros(rows, W, dtype=L.elem(), device=self.device)
        x = Act(self.x_rows.view(torch.uint8).view(-1), rows, W)
        for i in range(o.layers - self.layer_idx):
            x = self._block(f"transformer.resblocks.{i}.", x, B, T, o.heads, 64, 64, True, release_x=i > 0)
            if self.taps is not None:
                self.taps[f"resblocks.{i}"] = x
        self.out = self._ln("ln_final", x, "ln_final")

    # ------------------------------------------------------------------ run
    @torch.no_grad()
    def forward(self, tokens: torch.Tensor):
        """tokens: int [B, T] (``open_clip.tokenize``: <start> ... <end> then zero padding; <end> is the largest id, so
        ``argmax`` finds it — clip_embedder.py:199)."""
        o = self.o
        tokens = tokens.to(self.device).long()
        if tuple(tokens.shape) != (self.B, o.context_length):
            raise ValueError(f"tokens must be [{self.B}, {o.context_length}], got {tuple(tokens.shape)}")
        if int(tokens.min()) < 0 or int(tokens.max()) >= o.vocab_size:
            raise ValueError("token id outside the vocabulary")
        self.x_rows.copy_((self.tok[tokens] + self.pos).reshape(-1, o.width))           # (clip_embedder.py:193-194)
        self.S.run()
        x16 = self.out.tensor().view(self.B, o.context_length, o.width)
        eot = x16[torch.arange(self.B, device=self.device), tokens.argmax(dim=-1)].contiguous()       # [B, width]
        xt = torch.empty(self.B, o.embed_dim, dtype=torch.float32, device=self.device)
        E = ops.Stream(record=False)
        E.gemm(ops.gemm_params(self.B, self.wt["text_projection"].shape[0], ops.linear_segs([(eot.data_ptr(), o.width, o.width)]),
                               self.wt["text_projection"], xt.data_ptr(), o.embed_dim, out_fp32=True), "text_projection")
        return xt[:, : o.embed_dim], x16.float()

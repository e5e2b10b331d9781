"""ORACLE (test infrastructure, CPU, fp32) — restatement of the OpenCLIP text tower as VideoMV walks it.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; the product path
(``videomv_amd``) never does.

The walk is the reference's: ``FrozenOpenCLIPTtxtVisualEmbedder.encode_with_transformer`` and ``text_transformer_forward``
(tools/modules/clip_embedder.py:192-201, 217-225): token + positional embedding, every residual attention block except the last
``layer_idx``, ``ln_final``, ``xt = x[argmax token] @ text_projection``.  The blocks are ``open_clip``'s
``ResidualAttentionBlock`` (open_clip_torch, the version the reference's environment pins is not recorded in /root/reference —
``import open_clip`` at clip_embedder.py:6 is the only trace): ``x = x + attn(ln_1(x), mask)``, ``x = x + c_proj(gelu(c_fc(ln_2(x))))``
with ``nn.MultiheadAttention`` (fused in_proj [q; k; v], 1/sqrt(d) scaling, additive causal mask = -inf above the diagonal) and
``nn.GELU()`` (exact erf).  PARITY UNPINNED against the package (absent here, no golden vectors in the reference's tree);
``tests/test_clip_cpu.py`` pins this file against ``torch.nn.MultiheadAttention`` / ``nn.LayerNorm`` / ``nn.GELU`` assembled in
the same published structure.
"""
import torch
import torch.nn.functional as F


def text_tower(sd, tokens, width, heads, layers, layer_idx=1, taps=None):
    """sd: open_clip text-side state dict (fp32); tokens int64 [B, T] -> (xt [B, embed_dim], x [B, T, width])."""
    B, T = tokens.shape
    d = width // heads
    x = sd["token_embedding.weight"].float()[tokens] + sd["positional_embedding"].float()          # :193-194
    mask = torch.full((T, T), float("-inf")).triu_(1)                                              # open_clip build_attention_mask
    for i in range(layers - layer_idx):                                                            # :217-225
        p = f"transformer.resblocks.{i}."
        h = F.layer_norm(x, (width,), sd[p + "ln_1.weight"].float(), sd[p + "ln_1.bias"].float(), 1e-5)
        qkv = h @ sd[p + "attn.in_proj_weight"].float().t() + sd[p + "attn.in_proj_bias"].float()
        q, k, v = (t.view(B, T, heads, d).transpose(1, 2) for t in qkv.split(width, dim=-1))       # [B, heads, T, d]
        s = (q * d ** -0.5) @ k.transpose(-1, -2) + mask
        a = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, T, width)
        x = x + a @ sd[p + "attn.out_proj.weight"].float().t() + sd[p + "attn.out_proj.bias"].float()
        h = F.layer_norm(x, (width,), sd[p + "ln_2.weight"].float(), sd[p + "ln_2.bias"].float(), 1e-5)
        h = F.gelu(h @ sd[p + "mlp.c_fc.weight"].float().t() + sd[p + "mlp.c_fc.bias"].float())
        x = x + h @ sd[p + "mlp.c_proj.weight"].float().t() + sd[p + "mlp.c_proj.bias"].float()
        if taps is not None:
            taps[f"resblocks.{i}"] = x
    x = F.layer_norm(x, (width,), sd["ln_final.weight"].float(), sd["ln_final.bias"].float(), 1e-5)   # :198
    xt = x[torch.arange(B), tokens.argmax(dim=-1)] @ sd["text_projection"].float()                     # :199
    return xt, x

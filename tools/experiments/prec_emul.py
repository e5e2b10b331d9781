"""Scratch: CPU emulation of the engine's storage roundings, to choose the storage format.

Every point where the HIP plan writes an activation to HBM is a `q()` (operand storage) or `qs()` (residual-stream
carrier storage); weights are rounded with `qw()`.  Modes: bf16 everywhere (round 1), fp16 everywhere, bf16 operands with
an fp32 residual stream.  Prints rel-L2(eps) and per-block rel-L2 against the fp32 oracle.
"""
import dataclasses
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from oracle.unet_ref import UNetCfg, unet_forward, block_plan, sinusoidal_embedding, _mlp  # noqa: E402
from oracle.weights import random_state_dict, unet_param_shapes  # noqa: E402


class Mode:
    def __init__(self, op, stream, p=None):
        self.op, self.stream, self.p = op, stream, p or op

    def q(self, x):
        return x.to(self.op).float() if self.op is not None else x

    def qs(self, x):
        return x.to(self.stream).float() if self.stream is not None else x

    def qp(self, x):
        return x.to(self.p).float() if self.p is not None else x


def attention(M, sd, p, xn, context, heads):
    ctx = xn if context is None else M.q(context)
    q = M.q(F.linear(xn, sd[f"{p}.to_q.weight"]))
    k = M.q(F.linear(ctx, sd[f"{p}.to_k.weight"]))
    v = M.q(F.linear(ctx, sd[f"{p}.to_v.weight"]))
    b, n, inner = q.shape
    dh = inner // heads
    sp = lambda t: t.reshape(b, t.shape[1], heads, dh).permute(0, 2, 1, 3)
    q, k, v = sp(q), sp(k), sp(v)
    s = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * dh ** -0.5, dim=-1)
    # P is rounded to the MFMA operand type un-normalised (exp(s - max)), emulate as relative rounding of P
    o = M.q(torch.matmul(M.qp(s), v))
    o = o.permute(0, 2, 1, 3).reshape(b, n, inner)
    return F.linear(o, sd[f"{p}.to_out.0.weight"], sd[f"{p}.to_out.0.bias"])


def ln_folded(M, sd, p, name, x):
    """LayerNorm folded into the consumer: the GEMM operand is the (operand-rounded) raw x; statistics in fp32."""
    xo = M.q(x)
    mean = x.mean(-1, keepdim=True)
    rstd = torch.rsqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-5)
    return (xo - mean) * rstd * sd[f"{p}.{name}.weight"] + sd[f"{p}.{name}.bias"]


def tblock(M, sd, p, x, context, heads):
    x = M.qs(attention(M, sd, f"{p}.attn1", ln_folded(M, sd, p, "norm1", x), None, heads) + x)
    x = M.qs(attention(M, sd, f"{p}.attn2", ln_folded(M, sd, p, "norm2", x), context, heads) + x)
    h = F.linear(ln_folded(M, sd, p, "norm3", x), sd[f"{p}.ff.net.0.proj.weight"], sd[f"{p}.ff.net.0.proj.bias"])
    a, g = h.chunk(2, dim=-1)
    h = M.q(a * F.gelu(g))
    h = F.linear(h, sd[f"{p}.ff.net.2.weight"], sd[f"{p}.ff.net.2.bias"])
    return M.qs(h + x)


def st(M, sd, p, x, context, heads):
    n, c, h, w = x.shape
    x_in = x
    x = M.q(F.group_norm(x, 32, sd[f"{p}.norm.weight"], sd[f"{p}.norm.bias"], 1e-6))
    x = x.permute(0, 2, 3, 1).reshape(n, h * w, c)
    x = M.qs(F.linear(x, sd[f"{p}.proj_in.weight"], sd[f"{p}.proj_in.bias"]))
    x = tblock(M, sd, f"{p}.transformer_blocks.0", x, context, heads)
    x = F.linear(M.q(x), sd[f"{p}.proj_out.weight"], sd[f"{p}.proj_out.bias"])
    x = x.reshape(n, h, w, c).permute(0, 3, 1, 2)
    return M.qs(x + x_in)


def tt(M, sd, p, x, heads):
    b, c, f, h, w = x.shape
    x_in = x
    x = M.q(F.group_norm(x, 32, sd[f"{p}.norm.weight"], sd[f"{p}.norm.bias"], 1e-6))
    x = x.permute(0, 3, 4, 2, 1).reshape(b * h * w, f, c)
    wi = sd[f"{p}.proj_in.weight"]
    x = M.qs(F.linear(x, wi.reshape(wi.shape[0], wi.shape[1]), sd[f"{p}.proj_in.bias"]))
    x = tblock(M, sd, f"{p}.transformer_blocks.0", x, None, heads)
    wo = sd[f"{p}.proj_out.weight"]
    x = F.linear(M.q(x), wo.reshape(wo.shape[0], wo.shape[1]), sd[f"{p}.proj_out.bias"])
    x = x.reshape(b, h, w, f, c).permute(0, 4, 3, 1, 2)
    return M.qs(x + x_in)


def tconv(M, sd, p, x):
    idn = x
    for i, (name, ci) in enumerate((("conv1", 2), ("conv2", 3), ("conv3", 3), ("conv4", 3))):
        x = M.q(F.silu(F.group_norm(x, 32, sd[f"{p}.{name}.0.weight"], sd[f"{p}.{name}.0.bias"], 1e-5)))
        x = F.conv3d(x, sd[f"{p}.{name}.{ci}.weight"], sd[f"{p}.{name}.{ci}.bias"], padding=(1, 0, 0))
        if i < 3:
            x = M.q(x)
    return M.qs(idn + x)


def res(M, sd, p, x, emb, batch):
    h = M.q(F.silu(F.group_norm(x, 32, sd[f"{p}.in_layers.0.weight"], sd[f"{p}.in_layers.0.bias"], 1e-5)))
    h = F.conv2d(h, sd[f"{p}.in_layers.2.weight"], sd[f"{p}.in_layers.2.bias"], padding=1)
    e = F.linear(M.q(F.silu(emb)), sd[f"{p}.emb_layers.1.weight"], sd[f"{p}.emb_layers.1.bias"])
    h = M.q(h + e[:, :, None, None])
    h = M.q(F.silu(F.group_norm(h, 32, sd[f"{p}.out_layers.0.weight"], sd[f"{p}.out_layers.0.bias"], 1e-5)))
    h = F.conv2d(h, sd[f"{p}.out_layers.3.weight"], sd[f"{p}.out_layers.3.bias"], padding=1)
    if f"{p}.skip_connection.weight" in sd:
        x = F.conv2d(M.q(x), sd[f"{p}.skip_connection.weight"], sd[f"{p}.skip_connection.bias"])
    h = M.qs(x + h)
    n, c, hh, ww = h.shape
    h5 = h.reshape(batch, n // batch, c, hh, ww).permute(0, 2, 1, 3, 4)
    h5 = tconv(M, sd, f"{p}.temopral_conv", h5)
    return h5.permute(0, 2, 1, 3, 4).reshape(n, c, hh, ww)


def run_block(M, sd, blk, x, emb, context, batch):
    for kind, p, m in blk:
        if kind == "conv_in":
            x = M.qs(F.conv2d(M.q(x), sd[f"{p}.weight"], sd[f"{p}.bias"], padding=1))
        elif kind == "res":
            x = res(M, sd, p, x, emb, batch)
        elif kind == "st":
            x = st(M, sd, p, x, context, m["heads"])
        elif kind == "tt":
            n, c, h, w = x.shape
            x5 = x.reshape(batch, n // batch, c, h, w).permute(0, 2, 1, 3, 4)
            x = tt(M, sd, p, x5, m["heads"]).permute(0, 2, 1, 3, 4).reshape(n, c, h, w)
        elif kind == "down":
            x = M.qs(F.conv2d(M.q(x), sd[f"{p}.op.weight"], sd[f"{p}.op.bias"], stride=2, padding=1))
        elif kind == "up":
            x = M.qs(F.conv2d(F.interpolate(M.q(x), scale_factor=2, mode="nearest"), sd[f"{p}.conv.weight"], sd[f"{p}.conv.bias"], padding=1))
    return x


@torch.no_grad()
def forward(M, sd, cfg, x, t, y, cam, taps):
    sd = {k: (M.q(v) if v.dim() > 1 else v) for k, v in sd.items()}
    b, c, f, h, w = x.shape
    emb = _mlp(sd, "time_embed", M.q(sinusoidal_embedding(t, cfg.dim)))
    emb = emb.repeat_interleave(f, dim=0)
    emb = emb + _mlp(sd, "camera_embedding", M.q(cam.reshape(b * f, -1)))
    context = y.repeat_interleave(f, dim=0)
    x = x.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    inp, mid, outb = block_plan(cfg)
    xs = []
    for blk in inp:
        x = run_block(M, sd, blk, x, emb, context, b)
        xs.append(x)
        taps[blk[0][1]] = x
    x = run_block(M, sd, mid, x, emb, context, b)
    taps["middle_block"] = x
    for blk in outb:
        x = torch.cat([x, xs.pop()], dim=1)
        x = run_block(M, sd, blk, x, emb, context, b)
        taps[blk[0][1]] = x
    x = M.q(F.silu(F.group_norm(x, 32, sd["out.0.weight"], sd["out.0.bias"], 1e-5)))
    x = F.conv2d(x, sd["out.2.weight"], sd["out.2.bias"], padding=1)
    return x.reshape(b, f, cfg.out_dim, h, w).permute(0, 2, 1, 3, 4)


def rel(a, b):
    return float((a - b).norm() / b.norm())


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
    if which == "full":
        cfg = UNetCfg()
        B, F_, H, W = 1, 24, 8, 8
        seed = 5
    else:
        cfg = UNetCfg(dim=64, dim_mult=[1, 2], num_heads=2, num_res_blocks=1, attn_scales=[1.0, 0.5])
        B, F_, H, W = 2, 8, 16, 16
        seed = 99
    torch.set_num_threads(32)
    sd = random_state_dict(unet_param_shapes(cfg), seed)
    gen = torch.Generator().manual_seed(6)
    x = torch.randn(B, 4, F_, H, W, generator=gen)
    t = torch.tensor([601, 21][:B])
    y = torch.randn(B, 77, 1024, generator=gen)
    cam = torch.randn(B, F_, 16, generator=gen)
    tr = {}
    ref = unet_forward(sd, cfg, x, t, y, cam, taps=tr)
    for name, M in (("bf16", Mode(torch.bfloat16, torch.bfloat16)),
                    ("bf16 ops + fp32 stream", Mode(torch.bfloat16, None)),
                    ("bf16 ops + fp16 stream", Mode(torch.bfloat16, torch.float16)),
                    ("fp16", Mode(torch.float16, torch.float16))):
        tp = {}
        e = forward(M, sd, cfg, x, t, y, cam, tp)
        per = {k: round(rel(tp[k], tr[k]), 4) for k in tp}
        print(f"{name:26s} eps {rel(e, ref):.4e}  max-block {max(per.values()):.4e}  blocks {list(per.values())}", flush=True)

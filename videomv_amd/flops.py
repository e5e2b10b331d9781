"""Algorithmic work of a recorded plan: 2*M*N*K per GEMM-shaped launch and 4*Nq*Nk*64 per attention problem/head
(QK^T + PV).  Matches torch.utils.flop_counter on the reference modules (7.336 TFLOP at 24x32x32,
18.885 TFLOP at 24x40x64 per forward) up to the K zero-padding of the two 4-channel latent convs, which is excluded."""
from . import _lib as L


def gemm_flops(p) -> float:
    f = 2.0 * p.M * p.N * p.ktot
    if p.epilogue == L.EPI_TATTN:      # fused q | k | v + temporal attention: + QK^T and PV of every (pixel, head): 4 * F * F * 64 each
        f += 4.0 * p.M * p.F * 64 * (p.N // 192)
    return f


def gemm_bytes(p) -> float:
    """Algorithmic (compulsory) HBM bytes of one implicit-GEMM launch: every distinct source read once (an im2col tap
    or a temporal shift re-reads rows that are already counted), the weights once, the residual once, the output once."""
    seen, b = set(), 0.0
    for i in range(p.nseg):
        sg = p.seg[i]
        if sg.src in seen:
            continue
        seen.add(sg.src)
        rows = p.M
        if sg.mode == L.SEG_SPATIAL and p.OH > 0:
            rows = (p.M // (p.OH * p.OW)) * p.IH * p.IW
        b += 2.0 * rows * sg.k
    n_out = p.N // 2 if p.epilogue == L.EPI_GEGLU else (p.N // 3 if p.epilogue == L.EPI_TATTN else p.N)
    b += 2.0 * p.N * p.ktot
    b += (4.0 if p.out_fp32 else 2.0) * p.M * n_out
    if p.residual:
        b += 2.0 * p.M * n_out
    return b


def attn_flops(p) -> float:
    return 4.0 * p.n_outer * p.heads * p.Nq * p.Nk * 64


def plan_flops(recorded):
    """-> dict(total, gemm, attention) in FLOP for a list of (op, params)."""
    g = a = 0.0
    for op, p in recorded:
        if op == L.OP_GEMM:
            g += gemm_flops(p)
        elif op == L.OP_ATTENTION:
            a += attn_flops(p)
    return dict(total=g + a, gemm=g, attention=a)


# reference-counted figures (SURVEY.md §8d / BASELINE.md §2), TFLOP per UNet forward at F = 24
UNET_FWD_TFLOP = {(32, 32): 7.336, (40, 64): 18.885}
VAE_DECODE_TFLOP_PER_FRAME = {(32, 32): 0.622, (40, 64): 1.564}

"""Weight repacking: reference state-dict tensors (fp32, PyTorch layouts) -> the 16-bit (``_lib.elem()``: fp16 | bf16) ``[N][K]`` matrices the
implicit-GEMM kernel consumes (``include/vmv.h``).  Done once at load time on the host/device with torch
tensor ops (pure data movement — no arithmetic of the hot path happens here).

K ordering rules (must match ``ops.conv3x3_segs`` / ``ops.temporal_segs``):
  * 3x3 conv  [N, C, 3, 3]     -> [N, 9*C']  tap-major (dy, dx row-major), channels within a tap;
                                  C' = C zero-padded to a multiple of 8 (only the 4-channel latent convs)
  * temporal  [N, C, 3, 1, 1]  -> [N, 3*C]   taps dt = -1, 0, +1
  * linear / 1x1 / Conv1d(k=1) -> [N, K]
  * GEGLU     [2*I, K]         -> rows interleaved in 16-row blocks: x[16j:16j+16], gate[16j:16j+16]
N is zero-padded to a multiple of 4 (the epilogue stores 4 channels per lane).
"""
import torch

from . import _lib as L


def _pad_rows(w: torch.Tensor, mult=4) -> torch.Tensor:
    n = w.shape[0]
    pad = (-n) % mult
    if pad:
        w = torch.cat([w, w.new_zeros((pad,) + tuple(w.shape[1:]))], dim=0)
    return w


def pack_linear(w: torch.Tensor, device) -> torch.Tensor:
    w = w.reshape(w.shape[0], -1)
    k = w.shape[1]
    padk = (-k) % 8
    if padk:
        w = torch.cat([w, w.new_zeros(w.shape[0], padk)], dim=1)
    return _pad_rows(w).to(device=device, dtype=L.elem()).contiguous()


def pack_conv3x3(w: torch.Tensor, device) -> torch.Tensor:
    n, c, kh, kw = w.shape
    assert kh == 3 and kw == 3
    padc = (-c) % 8
    if padc:
        w = torch.cat([w, w.new_zeros(n, padc, 3, 3)], dim=1)
    w = w.permute(0, 2, 3, 1).reshape(n, -1)
    return _pad_rows(w).to(device=device, dtype=L.elem()).contiguous()


def pack_tconv(w: torch.Tensor, device) -> torch.Tensor:
    n, c, kt, kh, kw = w.shape
    assert kt == 3 and kh == 1 and kw == 1
    w = w.reshape(n, c, 3).permute(0, 2, 1).reshape(n, 3 * c)
    return _pad_rows(w).to(device=device, dtype=L.elem()).contiguous()


def geglu_interleave(t: torch.Tensor) -> torch.Tensor:
    """[2*I, ...] (x rows then gate rows, ``chunk(2)`` order of util.py:548) -> 16-row interleaved."""
    two_i = t.shape[0]
    i = two_i // 2
    assert i % 16 == 0
    x, g = t[:i], t[i:]
    rest = tuple(t.shape[1:])
    x = x.reshape((i // 16, 1, 16) + rest)
    g = g.reshape((i // 16, 1, 16) + rest)
    return torch.cat([x, g], dim=1).reshape((two_i,) + rest)


# position -> channel inside every block of 32 hidden channels: the order in which gemm_ff.hip's FF1 MFMA outputs (two 16-channel
# tiles, lane group g holding channels 4g..4g+3 of each) form the B operand of its FF2 MFMAs (include/vmv.h, VmvFfParams.w2)
FF_DOWN_ORDER = [4 * g + e if e < 4 else 16 + 4 * g + (e - 4) for g in range(4) for e in range(8)]


def ff_down_permute(w2: torch.Tensor) -> torch.Tensor:
    """FF down-projection weight [C, 4C] (util.py:573) with its K axis reordered per 32-channel block for the fused FeedForward."""
    n, k = w2.shape
    assert k % 32 == 0
    idx = torch.tensor(FF_DOWN_ORDER, dtype=torch.long)
    return w2.reshape(n, k // 32, 32)[:, :, idx].reshape(n, k).contiguous()


def fold_layernorm(w: torch.Tensor, b, gamma: torch.Tensor, beta: torch.Tensor):
    """LayerNorm folded into the Linear that consumes it (include/vmv.h, VmvGemmParams.rowstat): for
    y = W LN(x) + b, LN(x) = (x - mean) * rstd * gamma + beta, returns (W', b', colsum) with W' = W diag(gamma) already
    rounded to the element type (what the GEMM multiplies), b' = b + W beta and colsum[n] = sum_k W'[n][k] of the ROUNDED W' (so that
    the epilogue's  rstd * (acc - mean * colsum)  cancels exactly what the MFMAs accumulated)."""
    w = w.reshape(w.shape[0], -1).float()
    wf = (w * gamma.float()[None, :]).to(L.elem())
    bf = w @ beta.float()
    if b is not None:
        bf = bf + b.float()
    return wf, bf, wf.float().sum(dim=1)


def pack_bias(b: torch.Tensor, device, n_pad_to=4) -> torch.Tensor:
    b = b.reshape(-1).float()
    pad = (-b.shape[0]) % n_pad_to
    if pad:
        b = torch.cat([b, b.new_zeros(pad)])
    return b.to(device).contiguous()


def f32(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().float().to(device).contiguous()


def check_finite_weights(w: dict, what: str):
    """One device-side reduction + ONE host sync over all packed tensors of an engine: a checkpoint holding inf / NaN (or values the
    16-bit pack turned into inf) is refused at load time.  With MODE.FP16_OVFL set the fp16 MFMA silently treats a NaN operand as 0
    (csrc/common.h, diffusion_ddim._check_finite), so this is where such a weight can still be seen.  VMV_CHECK_FINITE=0 disables."""
    import os
    if os.environ.get("VMV_CHECK_FINITE", "1") == "0":
        return
    ts = [t for t in w.values() if torch.is_tensor(t) and t.is_floating_point() and t.is_cuda]
    if not ts:
        return
    ok = torch.stack([torch.isfinite(t).all() for t in ts])
    if not bool(ok.all()):
        names = [k for (k, t), good in zip([(k, t) for k, t in w.items() if torch.is_tensor(t) and t.is_floating_point() and t.is_cuda], ok.tolist()) if not good]
        raise FloatingPointError(f"{what}: packed weights hold inf / NaN ({', '.join(names[:5])}{' ...' if len(names) > 5 else ''}) — "
                                 f"values beyond the {L.elem_name()} range?" + (" Use hip_dtype: bf16 (VMV_DTYPE=bf16)." if L.elem_name() == "fp16" else ""))


def qkv_head_major(t: torch.Tensor, head_dim: int = 64) -> torch.Tensor:
    """Rows (or entries) of a fused [q | k | v] projection re-ordered head-major: [head][q | k | v][head_dim] — the order in which the
    fused projection + temporal-attention kernel (csrc/gemm_tqa.hip, VMV_EPI_TATTN) streams the weight matrix: all of one head's
    q, k and v rows arrive together and the head's attention is finished before the next head's rows.  Works on [3 C, K] matrices and
    on [3 C] vectors (bias, colsum)."""
    n3 = t.shape[0]
    inner = n3 // 3
    heads = inner // head_dim
    assert n3 == 3 * heads * head_dim
    idx = torch.arange(n3).view(3, heads, head_dim).permute(1, 0, 2).reshape(-1)
    return t[idx.to(t.device)]

#!/usr/bin/env python
"""Micro-benchmark of the attention kernels on the UNet's shapes at latent 24x40x64 (GPU only): TFLOP/s per shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videomv_amd import _lib as L, ops


def bench(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    S = ops.Stream(record=False)
    BF = L.elem()
    B, F = 2, 24
    shapes = [("self L0 N2560 h5", "spatial", 2560, 5), ("self L1 N640 h10", "spatial", 640, 10), ("self L2 N160 h20", "spatial", 160, 20),
              ("cross L0 N2560x77 h5", "cross", 2560, 5), ("cross L1 N640x77 h10", "cross", 640, 10),
              ("temporal L0 24 h5", "temporal", 2560, 5), ("temporal L1 24 h10", "temporal", 640, 10)]
    for name, kind, HW, heads in shapes:
        inner = heads * 64
        T = B * F * HW
        sc = 64 ** -0.5
        o = torch.zeros(T, inner, dtype=BF, device="cuda")
        if kind == "cross":
            q = torch.randn(T, inner, device="cuda").to(BF)
            kv = torch.randn(B * 77, 2 * inner, device="cuda").to(BF)
            mp = ops.seq_map(HW * inner, 0, inner, inner=1)
            kvm = ops.seq_map(77 * 2 * inner, 0, 2 * inner, inner=1)
            p = ops.attn_params(q, kv, kv.data_ptr() + 2 * inner, o, mp, kvm, kvm, mp, B * F, heads, HW, 77, sc, kv_div=F)
            fl = 4.0 * B * F * heads * HW * 77 * 64
        else:
            qkv = torch.randn(T, 3 * inner, device="cuda").to(BF)
            ld = 3 * inner
            if kind == "temporal":
                mp = lambda l: ops.seq_map(F * HW * l, l, HW * l, inner=HW)
                n_outer, N = B * HW, F
            else:
                mp = lambda l: ops.seq_map(HW * l, 0, l, inner=1)
                n_outer, N = B * F, HW
            base = qkv.data_ptr()
            p = ops.attn_params(base, base + 2 * inner, base + 4 * inner, o, mp(ld), mp(ld), mp(ld), mp(inner), n_outer, heads, N, N, sc)
            fl = 4.0 * n_outer * heads * N * N * 64
        ms = bench(lambda: S.attention(p))
        print(f"{name:24s} {ms * 1000:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Fused FeedForward (gemm_ff.hip) against the two-GEMM form at the UNet's L0 shape (M = 122 880, C = 320): us per FF."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videomv_amd import _lib as L, ops, packing as P

BF = L.elem()


def bench(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1000


for M in (122880, 61440):
    Cc = 320
    dev = "cuda"
    x = torch.randn(M, Cc, device=dev).to(BF)
    w1 = (torch.randn(8 * Cc, Cc, device=dev) * Cc ** -0.5).to(BF)
    b1 = torch.randn(8 * Cc, device=dev)
    cs = torch.randn(8 * Cc, device=dev)
    w2 = (torch.randn(Cc, 4 * Cc, device=dev) * (4 * Cc) ** -0.5).to(BF)
    b2 = torch.randn(Cc, device=dev)
    hid = torch.empty(M, 4 * Cc, device=dev, dtype=BF)
    out = torch.empty(M, Cc, device=dev, dtype=BF)
    S = ops.Stream(record=False)
    p1 = ops.gemm_params(M, 8 * Cc, ops.linear_segs([(x, Cc, Cc)]), w1, hid, 4 * Cc, bias=b1, colsum=cs, ln_eps=1e-5, epilogue=L.EPI_GEGLU)
    p2 = ops.gemm_params(M, Cc, ops.linear_segs([(hid, 4 * Cc, 4 * Cc)]), w2, out, Cc, bias=b2, residual=x, ldr=Cc)
    pf = ops.ff_params(M, Cc, x, Cc, w1, b1, w2, b2, out, Cc, residual=x, ldr=Cc, ln_eps=1e-5)
    t1, t2 = bench(lambda: S.gemm(p1)), bench(lambda: S.gemm(p2))
    tf = bench(lambda: S.ff(pf))
    fl = 2.0 * M * Cc * 12 * Cc
    print(f"M={M}: geglu {t1:7.1f} us + down {t2:7.1f} us = {t1 + t2:7.1f} us ({fl / (t1 + t2) / 1e6:6.1f} TFLOP/s)   fused {tf:7.1f} us ({fl / tf / 1e6:6.1f} TFLOP/s)", flush=True)

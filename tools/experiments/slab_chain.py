#!/usr/bin/env python
"""Experiment: does running a row-local chain of launches slab by slab keep its intermediates in the 256-MB Infinity Cache?
The FeedForward of an L0 transformer block writes a [M, 4C] hidden tensor (315 MB at M = 122 880, C = 320) with one launch and
re-reads it with the next.  Row-local ops can be replayed on row slabs (pointer offsets only — no new kernel): slab s runs GEGLU then
FF-down before slab s + 1 starts, so the hidden slab (315 MB / n) is re-read while it is still cache-resident.
    python tools/experiments/slab_chain.py      -> us per (GEGLU + down) for n = 1, 2, 4, 8, 16 slabs, L0 and L1 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videomv_amd import _lib as L, ops

BF = L.elem()


def bench(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1000


def main():
    dev = "cuda"
    S = ops.Stream(record=False)
    for name, M, C in (("L0", 122880, 320), ("L1", 30720, 640)):
        x = torch.randn(M, C, device=dev).to(BF)
        w1 = (torch.randn(8 * C, C, device=dev) * C ** -0.5).to(BF)
        b1 = torch.randn(8 * C, device=dev)
        cs = torch.randn(8 * C, device=dev)
        w2 = (torch.randn(C, 4 * C, device=dev) * (4 * C) ** -0.5).to(BF)
        b2 = torch.randn(C, device=dev)
        hid = torch.empty(M, 4 * C, device=dev, dtype=BF)
        out = torch.empty(M, C, device=dev, dtype=BF)
        line = f"{name} M={M} C={C}: "
        for n in (1, 2, 3, 4, 6, 8, 12, 16):
            if M % (n * 512):
                continue
            ms = M // n
            plist = []
            for s in range(n):
                xo, ho, oo = x.data_ptr() + 2 * s * ms * C, hid.data_ptr() + 2 * s * ms * 4 * C, out.data_ptr() + 2 * s * ms * C
                p1 = ops.gemm_params(ms, 8 * C, ops.linear_segs([(xo, C, C)]), w1, ho, 4 * C, bias=b1, epilogue=L.EPI_GEGLU, colsum=cs, ln_eps=1e-5)
                p2 = ops.gemm_params(ms, C, ops.linear_segs([(ho, 4 * C, 4 * C)]), w2, oo, C, bias=b2, residual=xo, ldr=C)
                plist.append((p1, p2))

            def run():
                for p1, p2 in plist:
                    S.gemm(p1); S.gemm(p2)
            line += f" n={n}:{bench(run):7.1f}us"
        print(line, flush=True)
        # the attention-side chain of a spatial block: out-proj(+res) -> LN q-proj -> (cross-attn omitted) -> out-proj(+res)
        w = (torch.randn(C, C, device=dev) * C ** -0.5).to(BF)
        a, b_, c_ = (torch.empty(M, C, device=dev, dtype=BF) for _ in range(3))
        line = f"{name} 3 x (K = N = C linear + residual) chain: "
        for n in (1, 2, 4, 8, 16):
            if M % (n * 512):
                continue
            ms = M // n
            plist = []
            for s in range(n):
                o = 2 * s * ms * C
                plist.append([ops.gemm_params(ms, C, ops.linear_segs([(src.data_ptr() + o, C, C)]), w, dst.data_ptr() + o, C, bias=b2, residual=x.data_ptr() + o, ldr=C)
                              for src, dst in ((x, a), (a, b_), (b_, c_))])

            def run():
                for ps in plist:
                    for p in ps:
                        S.gemm(p)
            line += f" n={n}:{bench(run):7.1f}us"
        print(line, flush=True)


if __name__ == "__main__":
    main()

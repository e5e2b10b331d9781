#!/usr/bin/env python
"""Vendor-library yardstick on the SAME box: hipBLASLt / rocBLAS (`F.linear`), MIOpen (`F.conv2d`, `F.conv3d`)
and PyTorch's SDPA timed on the SURVEY App. B shapes next to this repo's hand-written kernels (dispatcher's choice, tile 0).

    python tools/vendor_yardstick.py [--out profiles/r4_vendor_yardstick.tsv]

Not a target and not on the product path (nothing in videomv_amd/ calls these libraries): the only same-node evidence of what
this silicon gives a tuned vendor kernel on exactly these shapes.  The vendor side computes the bare contraction (no bias /
residual / GEGLU / LayerNorm fold / fused im2col-of-concat epilogues the hand kernels carry), fp16 in / fp16 out, fp32 accumulate;
convolutions run channels-last (the layout this repo keeps) and channels-first (MIOpen's native one) and the better is reported.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from videomv_amd import _lib as L, ops, packing as P

BF = L.elem()


def bench(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    torch.backends.cudnn.benchmark = True          # MIOpen find mode: let it pick its best solver per shape
    dev = "cuda"
    S = ops.Stream(record=False)
    M0, M1, M2, M3 = 122880, 30720, 7680, 1920
    rows = []

    def emit(name, flops, ms_hand, ms_vendor, vendor):
        th, tv = flops / ms_hand / 1e9, flops / ms_vendor / 1e9
        rows.append((name, f"{flops / 1e9:.1f}", f"{1000 * ms_hand:.1f}", f"{th:.0f}", vendor, f"{1000 * ms_vendor:.1f}", f"{tv:.0f}", f"{th / tv:.2f}"))
        print(f"{name:34s} hand {1000 * ms_hand:8.1f} us {th:7.0f} TF/s | {vendor:22s} {1000 * ms_vendor:8.1f} us {tv:7.0f} TF/s | hand/vendor {th / tv:5.2f}", flush=True)

    # ---- linears (App. B: qkv, out-proj / proj, GEGLU up (as a plain N = 8C GEMM for the vendor), FF down)
    for name, M, N, K in (("qkv L0", M0, 960, 320), ("out/proj L0", M0, 320, 320), ("geglu-up L0", M0, 2560, 320), ("ff-down L0", M0, 320, 1280),
                          ("qkv L1", M1, 1920, 640), ("out/proj L1", M1, 640, 640), ("geglu-up L1", M1, 5120, 640), ("ff-down L1", M1, 640, 2560),
                          ("qkv L2", M2, 3840, 1280), ("out/proj L2", M2, 1280, 1280), ("geglu-up L2", M2, 10240, 1280), ("ff-down L2", M2, 1280, 5120),
                          ("out/proj L3", M3, 1280, 1280)):
        x = torch.randn(M, K, device=dev).to(BF)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF)
        out = torch.empty(M, N, device=dev, dtype=BF)
        p = ops.gemm_params(M, N, ops.linear_segs([(x, K, K)]), w, out, N)
        emit(f"linear {name} {M}x{N}x{K}", 2.0 * M * N * K, bench(lambda: S.gemm(p)), bench(lambda: F.linear(x, w)), "F.linear (hipBLASLt)")
        del x, w, out

    # ---- 3x3 convolutions (stride 1, pad 1), 48 images (cond + uncond x 24 frames)
    for name, C, N, h, w_ in (("conv L0 320->320", 320, 320, 40, 64), ("conv L1 640->640", 640, 640, 20, 32), ("conv L2 1280->1280", 1280, 1280, 10, 16),
                              ("conv L3 1280->1280", 1280, 1280, 5, 8), ("conv dec L0 960->320", 960, 320, 40, 64)):
        nimg = 48
        M = nimg * h * w_
        xr = torch.randn(M, C, device=dev).to(BF)
        wt = (torch.randn(N, C, 3, 3, device=dev) * (9 * C) ** -0.5).to(BF)
        wp = P.pack_conv3x3(wt.float().cpu(), torch.device(dev))
        out = torch.empty(M, N, device=dev, dtype=BF)
        g = ops.Geom(OH=h, OW=w_, IH=h, IW=w_)
        ks, ws = ops.SplitK(torch.device(dev), cap=8).pick(M, N, ops.conv3x3_segs([(xr, C, C)]))
        p = ops.gemm_params(M, wp.shape[0], ops.conv3x3_segs([(xr, C, C)]), wp, out, N, geom=g, ksplit=ks, workspace=ws)
        ms_h = bench(lambda: S.gemm(p))
        x_cf = xr.view(nimg, h, w_, C).permute(0, 3, 1, 2).contiguous()                       # NCHW
        x_cl = x_cf.contiguous(memory_format=torch.channels_last)
        w_cl = wt.contiguous(memory_format=torch.channels_last)
        best = None
        for tag, xx, ww in (("NCHW", x_cf, wt), ("NHWC", x_cl, w_cl)):
            try:
                ms = bench(lambda: F.conv2d(xx, ww, padding=1), reps=6, warm=2)
                if best is None or ms < best[0]:
                    best = (ms, tag)
            except Exception as e:       # (a solver MIOpen cannot build offline)
                print("  conv2d", tag, "failed:", type(e).__name__, str(e)[:100])
        if best:
            emit(f"{name} @{h}x{w_}", 2.0 * M * N * 9 * C, ms_h, best[0], f"F.conv2d MIOpen {best[1]}")
        del xr, wt, wp, out, x_cf, x_cl, w_cl

    # ---- temporal (3,1,1) convolutions, zero-padded over 24 frames, 2 samples
    for name, C, hw in (("tconv L0 320", 320, 2560), ("tconv L1 640", 640, 640), ("tconv L2 1280", 1280, 160), ("tconv L3 1280", 1280, 40)):
        M = 2 * 24 * hw
        xr = torch.randn(M, C, device=dev).to(BF)
        wt = (torch.randn(C, C, 3, 1, 1, device=dev) * (3 * C) ** -0.5).to(BF)
        wp = P.pack_tconv(wt.float().cpu(), torch.device(dev))
        out = torch.empty(M, C, device=dev, dtype=BF)
        segs = ops.temporal_segs(xr, C, C)
        ks, ws = ops.SplitK(torch.device(dev), cap=8).pick(M, C, segs)
        p = ops.gemm_params(M, C, segs, wp, out, C, geom=ops.Geom(F=24, P=hw), ksplit=ks, workspace=ws)
        ms_h = bench(lambda: S.gemm(p))
        x5 = xr.view(2, 24, hw, 1, C).permute(0, 4, 1, 2, 3).contiguous()                     # [b, C, F, HW, 1]
        try:
            ms_v = bench(lambda: F.conv3d(x5, wt, padding=(1, 0, 0)), reps=6, warm=2)
            emit(f"{name} (M={M})", 2.0 * M * C * 3 * C, ms_h, ms_v, "F.conv3d MIOpen NCDHW")
        except Exception as e:
            print("  conv3d failed:", type(e).__name__, str(e)[:100])
        del xr, wt, wp, out, x5

    # ---- attention: spatial self (48 frames x heads, N = HW), cross (77 keys)
    for name, nb, heads, nq, nk in (("self-attn L0", 48, 5, 2560, 2560), ("self-attn L1", 48, 10, 640, 640), ("self-attn L2", 48, 20, 160, 160),
                                    ("cross-attn L0", 48, 5, 2560, 77)):
        inner = heads * 64
        q = torch.randn(nb * nq, inner, device=dev).to(BF)
        k = torch.randn(nb * nk, inner, device=dev).to(BF)
        v = torch.randn(nb * nk, inner, device=dev).to(BF)
        o = torch.empty(nb * nq, inner, device=dev, dtype=BF)
        qm, km = ops.seq_map(nq * inner, 0, inner, inner=1), ops.seq_map(nk * inner, 0, inner, inner=1)
        p = ops.attn_params(q, k, v, o, qm, km, km, qm, nb, heads, nq, nk, 0.125)
        ms_h = bench(lambda: S.attention(p))
        q4 = q.view(nb, nq, heads, 64).transpose(1, 2)
        k4 = k.view(nb, nk, heads, 64).transpose(1, 2)
        v4 = v.view(nb, nk, heads, 64).transpose(1, 2)
        try:
            ms_v = bench(lambda: F.scaled_dot_product_attention(q4, k4, v4), reps=6, warm=2)
            emit(f"{name} {nb}x{heads} Nq={nq} Nk={nk}", 4.0 * nb * heads * nq * nk * 64, ms_h, ms_v, "F.sdpa (torch)")
        except Exception as e:
            print("  sdpa failed:", type(e).__name__, str(e)[:100])
        del q, k, v, o

    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            f.write(f"# vendor yardstick, dtype {L.elem_name()}, torch {torch.__version__}, device {torch.cuda.get_device_name(0)}\n")
            f.write("shape\tGFLOP\thand_us\thand_TFLOPs\tvendor\tvendor_us\tvendor_TFLOPs\thand_over_vendor\n")
            for r in rows:
                f.write("\t".join(r) + "\n")


if __name__ == "__main__":
    main()

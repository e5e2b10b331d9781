#!/usr/bin/env python
"""Micro-benchmark of the implicit-GEMM kernels on the UNet's characteristic shapes (GPU only).
   python tools/gemm_bench.py [tile ...]   — prints TFLOP/s per (shape, tile); tile 0 = the dispatcher's choice."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from videomv_amd import _lib as L, ops

BF = L.elem()
N160 = (L.TILE_128x160, L.TILE_256x160, L.TILE_G128x160, L.TILE_P256x160, L.TILE_PP256x160, L.TILE_Q96x160, L.TILE_S192x160, L.TILE_S256x160, L.TILE_A128x160)


def bench(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    tiles = [int(t) for t in sys.argv[1:]] or [L.TILE_128x160, L.TILE_256x160]
    S = ops.Stream(record=False)
    dev = "cuda"
    M0, M1, M2 = 122880, 30720, 7680
    shapes = [  # (name, M, N, C(=K per tap), kind)
        ("lin+res L0 N320 K320", M0, 320, 320, "linres"), ("qkv  L0 N960 K320", M0, 960, 320, "lin"),
        ("lnqkv L0 N960 K320", M0, 960, 320, "lnlin"), ("lngeglu L0 N2560 K320", M0, 2560, 320, "lngeglu"),
        ("lnqkv L1 N1920 K640", M1, 1920, 640, "lnlin"), ("lngeglu L1 N5120 K640", M1, 5120, 640, "lngeglu"),
        ("geglu L0 N2560 K320", M0, 2560, 320, "geglu"), ("down L0 N320 K1280", M0, 320, 1280, "linres"),
        ("lin+res L1 N640 K640", M1, 640, 640, "linres"), ("qkv  L1 N1920 K640", M1, 1920, 640, "lin"),
        ("geglu L1 N5120 K640", M1, 5120, 640, "geglu"), ("down L1 N640 K2560", M1, 640, 2560, "linres"),
        ("lin+res L2 N1280 K1280", M2, 1280, 1280, "linres"), ("geglu L2 N10240 K1280", M2, 10240, 1280, "geglu"),
        ("conv L0 320->320", M0, 320, 320, "conv"), ("tcnv L0 320", M0, 320, 320, "tconv"),
        ("conv L1 640->640", M1, 640, 640, "conv"), ("tcnv L1 640", M1, 640, 640, "tconv"),
        ("conv L2 1280", M2, 1280, 1280, "conv"),
        ("conv L3 1280", 1920, 1280, 1280, "conv"), ("conv L3 2560->1280", 1920, 1280, 2560, "conv"),
        ("tcnv L3 1280", 1920, 1280, 1280, "tconv"), ("down L3 N1280 K5120", 1920, 1280, 5120, "linres"),
        ("lin+res L3 N1280 K1280", 1920, 1280, 1280, "linres"),
        # VAE decoder levels at 24 frames of 320 x 512 (decode_views takes all frames in one plan)
        ("vae conv 512 @80x128", 24 * 80 * 128, 512, 512, "conv"), ("vae conv 256 @160x256", 24 * 160 * 256, 256, 256, "conv"),
        ("vae conv 128 @320x512", 24 * 320 * 512, 128, 128, "conv"),
    ]
    flt = os.environ.get("VMV_BENCH_SHAPES", "")
    for name, M, N, C, kind in shapes:
        if flt and not any(f in name for f in flt.split(",")):
            continue
        x = torch.randn(M, C, device=dev).to(BF)
        kw = {}
        No = N
        if kind in ("lin", "linres", "geglu", "lnlin", "lngeglu"):
            K = C; segs = ops.linear_segs([(x, C, C)]); geom = None
            if kind in ("lnlin", "lngeglu"):
                kw["colsum"] = torch.randn(N, device=dev)
                kw["ln_eps"] = 1e-5      # (in-kernel statistics where the kernel takes them: gemm_rs; ignored next to a rowstat)
                if os.environ.get("VMV_BENCH_LN_INLINE", "0") != "1":     # 1: statistics in the persistent GEMM's own main loop
                    kw["rowstat"] = torch.randn(M, 2, device=dev).abs() + 0.5
            if kind in ("geglu", "lngeglu"):
                kw["epilogue"] = L.EPI_GEGLU; No = N // 2
        elif kind == "conv":
            K = 9 * C; segs = ops.conv3x3_segs([(x, C, C)])
            G = int(os.environ.get("VMV_BENCH_CGROUPS", "1"))       # experiment: channel-group-major K order (G groups x 9 taps)
            if G > 1 and C % (8 * G) == 0:
                cg = C // G
                segs = [ops.Seg(x.data_ptr() + 2 * g_ * cg, C, cg, L.SEG_SPATIAL, dy, dx) for g_ in range(G) for (dy, dx) in ops.TAPS3x3]
            hw = {M0: (40, 64), M1: (20, 32), M2: (10, 16), 1920: (5, 8), 24 * 80 * 128: (80, 128), 24 * 160 * 256: (160, 256),
                  24 * 320 * 512: (320, 512)}[M]
            geom = ops.Geom(OH=hw[0], OW=hw[1], IH=hw[0], IW=hw[1])
        else:
            K = 3 * C; segs = ops.temporal_segs(x, C, C); geom = ops.Geom(F=24, P=M // 48)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF)
        b = torch.randn(N, device=dev)
        out = torch.empty(M, No, device=dev, dtype=BF)
        if kind == "linres":
            res = torch.randn(M, No, device=dev).to(BF)
            kw.update(residual=res, ldr=No)
        line = f"{name:24s}"
        for tile in tiles:
            if tile in N160 and (N % 160 or kind in ("geglu", "lngeglu")):
                line += "      -    "; continue
            if tile in (L.TILE_X256x320, L.TILE_X256x256, L.TILE_X256x128) and (kind in ("geglu", "lngeglu", "lnlin") or (tile == L.TILE_X256x320 and N % 320)):
                line += "      -    "; continue
            if tile in (L.TILE_RS, L.TILE_RS512, L.TILE_RS256) and (kind in ("conv", "tconv") or C not in (320, 640) or (tile == L.TILE_RS512 and C != 320)):
                line += "      -    "; continue
            stamps = torch.zeros(8 * 64, dtype=torch.int64, device=dev) if os.environ.get("VMV_GEMM_ABLATE") in ("4", "7", "8") else None
            ks = int(os.environ.get("VMV_BENCH_KSPLIT", "0"))
            if ks > 1:
                stamps = torch.zeros(ks * M * N, device=dev)
                kw["ksplit"] = ks
            p = ops.gemm_params(M, N, segs, w, out, No, bias=b, geom=geom, tile=tile, workspace=stamps, **kw)
            ms = bench(lambda: S.gemm(p))
            line += f" t{tile}:{2.0 * M * N * K / ms / 1e9:7.1f}"
            if stamps is not None and tile in (L.TILE_P256x128, L.TILE_P256x160, L.TILE_Q128x128, L.TILE_Q96x160, L.TILE_PP256x128, L.TILE_PP256x160):
                t = stamps.cpu().view(-1, 4)
                t = t[t[:, 0] > 0]
                if tile in (L.TILE_PP256x128, L.TILE_PP256x160):
                    t = stamps.cpu().view(-1, 8)[:2]
                    base = int(t[0, 0])
                    line += "\n      PP stamps chunk 8 (grp0 / grp1: LOAD start, reads issued, DMA issued, waits done, barrier, MFMAs issued, wait, barrier):\n      " + \
                            "\n      ".join(" ".join(f"{int(v) - base:6d}" for v in r) for r in t)
                elif len(t):
                    base = int(t[0, 0])
                    rows = [" ".join(f"{int(v) - base:7d}" for v in r) for r in t[:6]]
                    line += "\n      stamps(block0; tile start / loop done / epi start / epi end):\n      " + "\n      ".join(rows)
            if stamps is not None and tile in (L.TILE_S256x128, L.TILE_S192x160, L.TILE_S256x160, L.TILE_A128x160):
                t = stamps.cpu().view(-1, 8)[:int(os.environ.get("VMV_STAMP_ROWS", "40"))]
                base = int(t[0, 0])
                line += "\n   chunk: loader[before wait, after wait, after B, after issue]  mfma[before B, after B]\n   " + \
                        "\n   ".join(f"{i:3d}: " + " ".join(f"{int(v) - base:7d}" for v in r[:6]) for i, r in enumerate(t))
        print(line, flush=True)


if __name__ == "__main__":
    main()

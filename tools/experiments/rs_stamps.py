#!/usr/bin/env python
"""Phase timeline of gemm_rs (a build with -DVMV_RS_ABLATE=8): block 0, waves 0 / 4, s_memtime at kernel start, rows resident,
first chunk visible, then per pair: start, MFMAs issued, epilogue issued (chunk end in between for the late half)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videomv_amd import _lib as L, ops

BF = L.elem()


def run(name, M, N, K, geglu=False, ln=False, res=False):
    dev = "cuda"
    x = torch.randn(M, K, device=dev).to(BF)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(BF)
    b = torch.randn(N, device=dev)
    No = N // 2 if geglu else N
    out = torch.empty(M, No, device=dev, dtype=BF)
    kw = {}
    if ln: kw.update(colsum=torch.randn(N, device=dev), ln_eps=1e-5)
    if geglu: kw["epilogue"] = L.EPI_GEGLU
    if res: kw.update(residual=torch.randn(M, No, device=dev).to(BF), ldr=No)
    stamps = torch.zeros(512, dtype=torch.int64, device=dev)
    p = ops.gemm_params(M, N, ops.linear_segs([(x, K, K)]), w, out, No, bias=b, tile=L.TILE_RS, workspace=stamps, **kw)
    S = ops.Stream(record=False)
    for _ in range(3):
        S.gemm(p)
    torch.cuda.synchronize()
    t = stamps.cpu().view(2, 256)
    print(f"== {name}: M={M} N={N} K={K}")
    for wv in range(2):
        r = t[wv]
        n = int((r > 0).sum())
        base = int(r[0])
        v = [int(x) - base for x in r[:n]]
        print(f"  wave {4 * wv}: start->rows {v[1]}  ->chunk0 {v[2] - v[1]}  loop {v[n - 1] - v[2]} cycles over {(n - 3) // 3} pairs; total {v[n - 1]}")
        pairs = [(v[3 + 3 * k + 1] - v[3 + 3 * k], v[3 + 3 * k + 2] - v[3 + 3 * k + 1], (v[3 + 3 * k + 3] - v[3 + 3 * k + 2]) if 3 + 3 * k + 3 < n else 0) for k in range((n - 3) // 3)]
        print("    (mfma, epilogue, gap to next pair) per pair:", " ".join(f"{a}/{b}/{c}" for a, b, c in pairs[:16]))
        if len(pairs) > 20:
            import statistics
            mid = pairs[4:-2]
            print("    steady state median:", statistics.median(p_[0] for p_ in mid), statistics.median(p_[1] for p_ in mid), statistics.median(p_[2] for p_ in mid))


if __name__ == "__main__":
    M0, M1 = 122880, 30720
    run("qkv L0", M0, 960, 320)
    run("lnqkv L0", M0, 960, 320, ln=True)
    run("lngeglu L0", M0, 2560, 320, geglu=True, ln=True)
    run("lin+res L0", M0, 320, 320, res=True)
    run("lngeglu L1", M1, 5120, 640, geglu=True, ln=True)
    run("qkv L1", M1, 1920, 640)

"""Experiment: does running the wave-specialised conv GEMM (tile 16) slow down the UNCHANGED kernels that follow it?
Times a geglu GEMM (tile 9) and an attention-free elementwise-ish op interleaved with a conv GEMM on tile 0 vs tile 16."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videomv_amd import _lib as L, ops
BF = torch.bfloat16
dev = "cuda"
S = ops.Stream(record=False)
M0 = 122880
x = torch.randn(M0, 320, device=dev).to(BF)
wc = (torch.randn(320, 2880, device=dev) * 2880 ** -0.5).to(BF); bc = torch.randn(320, device=dev)
oc = torch.empty(M0, 320, device=dev, dtype=BF)
wg = (torch.randn(2560, 320, device=dev) * 320 ** -0.5).to(BF); bg = torch.randn(2560, device=dev)
og = torch.empty(M0, 1280, device=dev, dtype=BF)
geom = ops.Geom(OH=40, OW=64, IH=40, IW=64)
def conv(tile): return ops.gemm_params(M0, 320, ops.conv3x3_segs([(x, 320, 320)]), wc, oc, 320, bias=bc, geom=geom, tile=tile)
pg = ops.gemm_params(M0, 2560, ops.linear_segs([(x, 320, 320)]), wg, og, 1280, bias=bg, epilogue=L.EPI_GEGLU, tile=9)
def run(conv_tile, reps=40):
    pc = conv(conv_tile) if conv_tile is not None else None
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(reps)]
    for _ in range(5):
        if pc: S.gemm(pc)
        S.gemm(pg)
    torch.cuda.synchronize()
    for r in range(reps):
        ev[r][0].record()
        if pc: S.gemm(pc)
        ev[r][1].record()
        S.gemm(pg)
        ev[r][2].record()
    torch.cuda.synchronize()
    tc = sum(e[0].elapsed_time(e[1]) for e in ev) / reps
    tg = sum(e[1].elapsed_time(e[2]) for e in ev) / reps
    return tc, tg
for rnd in range(2):
    for ct in (None, 0, 16, 6):
        tc, tg = run(ct)
        print(f"conv tile {ct}: conv {tc*1000:7.1f} us   geglu(t9) {tg*1000:7.1f} us   sum {1000*(tc+tg):7.1f}")

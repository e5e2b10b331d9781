"""Experiment: sustained (seconds) vs burst (milliseconds) rate of one GEMM, with the shader clock rocm-smi reports."""
import sys, os, subprocess, re, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videomv_amd import _lib as L, ops
BF = torch.bfloat16
dev = "cuda"
S = ops.Stream(record=False)
def sclk():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out); p = re.search(r"Power \(W\):\s*([\d.]+)", out)
    return (int(c.group(1)) if c else -1, float(p.group(1)) if p else -1)
def case(name, M, N, C, hw, tile):
    x = torch.randn(M, C, device=dev).to(BF)
    w = (torch.randn(N, 9 * C, device=dev) * (9 * C) ** -0.5).to(BF); b = torch.randn(N, device=dev)
    o = torch.empty(M, N, device=dev, dtype=BF)
    p = ops.gemm_params(M, N, ops.conv3x3_segs([(x, C, C)]), w, o, N, bias=b, geom=ops.Geom(OH=hw[0], OW=hw[1], IH=hw[0], IW=hw[1]), tile=tile)
    fl = 2.0 * M * N * 9 * C
    time.sleep(1.0)
    line = f"{name} tile {tile}: idle clk/power {sclk()} | TF/s per 0.4 s window:"
    for win in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 0
        t0 = time.perf_counter()
        e0.record()
        while time.perf_counter() - t0 < 0.4:
            for _ in range(20): S.gemm(p)
            n += 20
            torch.cuda.current_stream().synchronize() if n % 200 == 0 else None
        e1.record(); torch.cuda.synchronize()
        line += f" {fl * n / (e0.elapsed_time(e1) * 1e-3) / 1e12:6.0f}"
        if win in (3, 7): line += f" {sclk()}"
    print(line, flush=True)
case("conv L0 320", 122880, 320, 320, (40, 64), 0)
case("conv L0 320", 122880, 320, 320, (40, 64), 16)
case("conv L1 640", 30720, 640, 640, (20, 32), 6)
case("conv L1 640", 30720, 640, 640, (20, 32), 16)
case("conv L2 1280", 7680, 1280, 1280, (10, 16), 6)
case("conv L2 1280", 7680, 1280, 1280, (10, 16), 17)

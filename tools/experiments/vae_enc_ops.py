#!/usr/bin/env python
"""Per-launch times of the VAE ENCODER plan (24 images of 256 x 256 -> latent 32 x 32: the LGM branch's re-encode)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from videomv_amd import _lib as L
from videomv_amd.registry import AUTO_ENCODER
import videomv_amd.autoencoder  # noqa: F401
from bench import randomize_
from videomv_amd.flops import gemm_flops

dev = torch.device("cuda")
n, hw = int(os.environ.get("N", 24)), int(os.environ.get("HW", 256))
dd = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)
with torch.device(dev):
    vae = AUTO_ENCODER.build(dict(type="AutoencoderKL", ddconfig=dd, embed_dim=4))
randomize_(vae, 4321)
x = torch.randn(n, 3, hw, hw, device=dev)
vae.encode(x)
eng = [e for e in vae._engines.values() if type(e).__name__ == "VaeEncoderEngine"][0]
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): vae.encode(x)
e1.record(); torch.cuda.synchronize()
print(f"encode-{n} @{hw}: {e0.elapsed_time(e1) / 3:.2f} ms")
rec, labels = eng.S.recorded, eng.S.labels
nn = len(rec)
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(nn + 1)] for _ in range(3)]
for r in range(3):
    torch.cuda.synchronize(); ev[r][0].record()
    for i in range(nn):
        eng.S.run(i, i + 1); ev[r][i + 1].record()
torch.cuda.synchronize()
ms = [min(ev[r][i].elapsed_time(ev[r][i + 1]) for r in range(3)) for i in range(nn)]
for i, (op, p) in enumerate(rec):
    fl = gemm_flops(p) if op == L.OP_GEMM else 0.0
    extra = f"M={p.M} N={p.N} K={p.ktot} ks={p.ksplit}" if op == L.OP_GEMM else ""
    print(f"{i:3d} {labels[i]:44s} {ms[i] * 1000:8.1f} us {fl / 1e9:8.1f} GF {fl / (ms[i] * 1e-3) / 1e12 if fl else 0:7.1f} TF  {extra}")
print("serial total", round(sum(ms), 2), "ms")

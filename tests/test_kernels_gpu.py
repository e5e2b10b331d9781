"""GPU parity tests of every HIP kernel, called through the C ABI (ctypes) — `pytest -m gpu`.

Reference = plain PyTorch fp32 math of the same op on the SAME bf16-rounded inputs (tests/plan_interp.py, written
from include/vmv.h; its conv/attention semantics are themselves checked against torch.nn.functional in
tests/test_interp_cpu.py).  Tolerances (stated per test): outputs are bf16 (8 mantissa bits, half-ulp 2^-9 =
0.2 %) of fp32-accumulated results, so max-abs error <= 1 % of the output scale and rel-L2 <= 4e-3; fp32 outputs
rel-L2 <= 1e-3 (bf16 operands, fp32 accumulate in a different order).
"""
import ctypes as C
import math

import pytest
import torch

from videomv_amd import _lib as L
from videomv_amd import ops, packing as P
from tests import plan_interp as I

pytestmark = pytest.mark.gpu
BF = L.elem()
# fp16 carries 3 more significand bits than bf16: every storage-rounding tolerance below is stated for bf16 and tightened 8x for fp16
TS = 1.0 if BF == torch.bfloat16 else 0.125


def rel_l2(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12))


def check(out_gpu, out_ref, tol_l2=4e-3, tol_max=1e-2):
    a, b = out_gpu.float().cpu(), out_ref.float()
    assert torch.isfinite(a).all()
    scale = float(b.abs().max().clamp_min(1e-6))
    e_max = float((a - b).abs().max()) / scale
    e_l2 = rel_l2(a, b)
    assert e_l2 < tol_l2 * TS and e_max < tol_max * TS, (e_l2, e_max, TS)


def g(seed):
    return torch.Generator().manual_seed(seed)


def rnd(shape, seed, scale=1.0, dtype=BF):
    return (torch.randn(shape, generator=g(seed)) * scale).to(dtype)


class Case:
    """Holds CPU tensors; `.on(dev)` clones them to a device; builders take the dict of tensors."""

    def __init__(self, **tensors):
        self.t = tensors

    def on(self, dev):
        return {k: (v.clone().to(dev).contiguous() if isinstance(v, torch.Tensor) else v) for k, v in self.t.items()}


def run_gemm(build, case, out_keys=("out",), cpu_ref=True):
    cpu = case.on("cpu")
    if cpu_ref:                       # (the big multi-tile cases check against torch ops instead of the slow interpreter)
        I.gemm(build(cpu))
    dev = case.on("cuda")
    S = ops.Stream(record=False)
    S.gemm(build(dev), "test")
    torch.cuda.synchronize()
    return cpu, dev


# ------------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K,tile", [(300, 320, 320, 0), (257, 128, 192, 0), (128, 64, 64, 0), (1000, 960, 320, 0),
                                        (77, 256, 1024, 0), (50, 4, 40, 0), (640, 640, 640, L.TILE_128x128),
                                        (640, 640, 640, L.TILE_128x160), (64, 64, 128, L.TILE_64x64),
                                        (640, 640, 640, L.TILE_256x128), (1000, 960, 320, L.TILE_256x160),
                                        (300, 320, 320, L.TILE_256x160), (257, 128, 192, L.TILE_256x128),
                                        (513, 200, 72, L.TILE_256x128), (2000, 1280, 1280, L.TILE_256x160),
                                        (640, 640, 640, L.TILE_G128x128), (1000, 960, 320, L.TILE_G128x160),
                                        (300, 320, 64, L.TILE_G128x160), (257, 128, 192, L.TILE_G128x128),
                                        (513, 200, 72, L.TILE_G128x128), (2000, 1280, 1280, L.TILE_G128x160),
                                        (640, 640, 640, L.TILE_P256x128), (1000, 960, 320, L.TILE_P256x160),
                                        (300, 320, 64, L.TILE_P256x160), (257, 128, 192, L.TILE_P256x128),
                                        (513, 200, 72, L.TILE_P256x128), (2000, 1280, 1280, L.TILE_P256x160),
                                        (70000, 320, 320, L.TILE_P256x160), (70000, 384, 128, L.TILE_P256x128),
                                        (640, 640, 640, L.TILE_W256x256), (2000, 1280, 1280, L.TILE_W256x256), (300, 320, 320, L.TILE_W256x256), (257, 128, 192, L.TILE_W256x256), (7000, 3840, 1280, L.TILE_W256x256), (513, 200, 160, L.TILE_W256x256),
                                        (66000, 160, 64, L.TILE_P256x160),
                                        (640, 640, 640, L.TILE_PP256x128), (1000, 960, 320, L.TILE_PP256x160),
                                        (300, 320, 64, L.TILE_PP256x160), (257, 128, 192, L.TILE_PP256x128),
                                        (513, 200, 72, L.TILE_PP256x128), (2000, 1280, 1280, L.TILE_PP256x160),
                                        (640, 640, 640, L.TILE_Q128x128), (1000, 960, 320, L.TILE_Q96x160),
                                        (300, 320, 64, L.TILE_Q96x160), (257, 128, 192, L.TILE_Q128x128),
                                        (513, 200, 72, L.TILE_Q128x128), (2000, 1280, 1280, L.TILE_Q96x160),
                                        (70000, 320, 320, L.TILE_Q96x160), (70000, 384, 128, L.TILE_Q128x128),
                                        (66000, 160, 64, L.TILE_Q96x160),
                                        (640, 640, 640, L.TILE_S256x128), (1000, 960, 320, L.TILE_S192x160),
                                        (300, 320, 64, L.TILE_S192x160), (257, 128, 192, L.TILE_S256x128),
                                        (513, 200, 72, L.TILE_S256x128), (2000, 1280, 1280, L.TILE_S192x160),
                                        (70000, 320, 320, L.TILE_S192x160), (70000, 384, 128, L.TILE_S256x128),
                                        (66000, 160, 64, L.TILE_S192x160), (1000, 960, 320, L.TILE_S256x160),
                                        (70000, 320, 320, L.TILE_S256x160), (2000, 1280, 1280, L.TILE_S256x160),
                                        (66000, 100, 72, L.TILE_S256x160),
                                        # wide-tile kernel (gemm_xglds.hip): M / N tails, several tiles per block column, K = 128 .. 1280
                                        (640, 640, 640, L.TILE_X256x320), (1000, 960, 320, L.TILE_X256x320), (300, 320, 128, L.TILE_X256x320),
                                        (2000, 1280, 1280, L.TILE_X256x320), (70000, 320, 320, L.TILE_X256x320), (513, 200, 192, L.TILE_X256x320),
                                        (640, 640, 640, L.TILE_X256x256), (257, 256, 192, L.TILE_X256x256), (2000, 1280, 1280, L.TILE_X256x256),
                                        (66000, 512, 128, L.TILE_X256x256), (640, 640, 640, L.TILE_X256x128), (257, 128, 192, L.TILE_X256x128),
                                        (70000, 384, 128, L.TILE_X256x128),
                                        # round 6: 256-thread blocks, two per CU (gemm_xglds.hip WNV = 4)
                                        (640, 640, 640, L.TILE_Y256x128), (257, 128, 192, L.TILE_Y256x128), (2000, 1280, 1280, L.TILE_Y256x128),
                                        (70000, 384, 128, L.TILE_Y256x128)])
def test_gemm_linear_bias(M, N, K, tile):
    c = Case(a=rnd((M, K), 1), w=rnd((N, K), 2, K ** -0.5), b=torch.randn(N, generator=g(3)), out=torch.zeros(M, N, dtype=BF))

    def build(t):
        return ops.gemm_params(M, N, ops.linear_segs([(t["a"], K, K)]), t["w"], t["out"], N, bias=t["b"], tile=tile)
    cpu, dev = run_gemm(build, c)
    check(dev["out"], cpu["out"])


@pytest.mark.parametrize("tile", [0, L.TILE_256x160, L.TILE_G128x160, L.TILE_P256x160, L.TILE_PP256x160, L.TILE_Q96x160, L.TILE_S192x160, L.TILE_S256x160])
def test_gemm_fp32_out_rowvec_act_residual(tile):
    M, N, K = 384, 320, 128
    c = Case(a=rnd((M, K), 1), w=rnd((N, K), 2, K ** -0.5), b=torch.randn(N, generator=g(3)),
             rv=torch.randn(M // 64, 512, generator=g(4)), res=rnd((M, N), 5), out=torch.zeros(M, N))

    def build(t):
        return ops.gemm_params(M, N, ops.linear_segs([(t["a"], K, K)]), t["w"], t["out"], N, bias=t["b"],
                               rowvec=t["rv"].data_ptr() + 4 * 64, rowvec_div=64, rowvec_ld=512, act=L.ACT_SILU,
                               residual=t["res"], ldr=N, out_fp32=True, tile=tile)
    cpu, dev = run_gemm(build, c)
    check(dev["out"], cpu["out"], tol_l2=1e-3, tol_max=2e-3)


@pytest.mark.parametrize("tile", [0, L.TILE_256x128, L.TILE_G128x128, L.TILE_P256x128, L.TILE_PP256x128, L.TILE_Q128x128, L.TILE_S256x128, L.TILE_X256x256, L.TILE_W256x256, L.TILE_Y256x128])
def test_gemm_geglu(tile):
    M, I2, K = 200, 512, 128        # 2*I = 512 rows -> 256 outputs
    w = rnd((I2, K), 2, K ** -0.5)
    b = torch.randn(I2, generator=g(3))
    c = Case(a=rnd((M, K), 1), w=P.geglu_interleave(w), b=P.geglu_interleave(b), out=torch.zeros(M, I2 // 2, dtype=BF))

    def build(t):
        return ops.gemm_params(M, I2, ops.linear_segs([(t["a"], K, K)]), t["w"], t["out"], I2 // 2, bias=t["b"],
                               epilogue=L.EPI_GEGLU, tile=tile)
    cpu, dev = run_gemm(build, c)
    check(dev["out"], cpu["out"])
    # and against the un-interleaved definition x * gelu(gate)  (util.py:548-550)
    h = cpu["a"].float() @ w.float().t() + b
    x, gate = h.chunk(2, dim=-1)
    check(dev["out"], x * torch.nn.functional.gelu(gate))


@pytest.mark.parametrize("tile", [0, L.TILE_256x128, L.TILE_G128x128, L.TILE_P256x128, L.TILE_PP256x128, L.TILE_Q128x128, L.TILE_S256x128, L.TILE_S192x160, L.TILE_S256x160,
                                  L.TILE_X256x320, L.TILE_X256x256, L.TILE_X256x128, L.TILE_X512x128, L.TILE_Y256x128])
@pytest.mark.parametrize("stride,ups,two_src,skip", [(1, 0, False, False), (2, 0, False, False), (1, 1, False, False),
                                                     (1, 0, True, True)])
def test_gemm_conv3x3(stride, ups, two_src, skip, tile):
    n, IH, IW, C0, C1, N = 3, 10, 12, 64, (32 if two_src else 0), 128
    OH = (IH + 1) // 2 if stride == 2 else (IH * 2 if ups else IH)
    OW = (IW + 1) // 2 if stride == 2 else (IW * 2 if ups else IW)
    Cin = C0 + C1
    wt = torch.randn(N, Cin, 3, 3, generator=g(2)) * (9 * Cin) ** -0.5
    ws = torch.randn(N, Cin, generator=g(7)) * Cin ** -0.5
    wp = wt.permute(0, 2, 3, 1).reshape(N, -1)
    if skip:
        wp = torch.cat([wp, ws], dim=1)
    c = Case(x0=rnd((n * IH * IW, C0), 1), x1=rnd((n * IH * IW, max(C1, 8)), 4), w=wp.to(BF),
             b=torch.randn(N, generator=g(3)), out=torch.zeros(n * OH * OW, N, dtype=BF))
    M = n * OH * OW

    def build(t):
        srcs = [(t["x0"], C0, C0)] + ([(t["x1"], max(C1, 8), C1)] if two_src else [])
        segs = ops.conv3x3_segs(srcs)
        if skip:
            segs += ops.linear_segs(srcs)
        return ops.gemm_params(M, N, segs, t["w"], t["out"], N, bias=t["b"],
                               geom=ops.Geom(OH=OH, OW=OW, IH=IH, IW=IW, stride=stride, ups=ups), tile=tile)
    cpu, dev = run_gemm(build, c)
    check(dev["out"], cpu["out"])
    # independent check against F.conv2d on the same rounded operands
    x = cpu["x0"].float()
    if two_src:
        x = torch.cat([x, cpu["x1"][:, :C1].float()], dim=1)
    img = x.view(n, IH, IW, Cin).permute(0, 3, 1, 2)
    if ups:
        img = torch.nn.functional.interpolate(img, scale_factor=2, mode="nearest")
    ref = torch.nn.functional.conv2d(img, wt.to(BF).float(), cpu["b"], stride=stride, padding=1)
    if skip:
        ref = ref + torch.nn.functional.conv2d(x.view(n, IH, IW, Cin).permute(0, 3, 1, 2), ws.to(BF).float()[:, :, None, None])
    check(dev["out"], ref.permute(0, 2, 3, 1).reshape(M, N))


@pytest.mark.parametrize("tile", [0, L.TILE_256x128, L.TILE_G128x128, L.TILE_P256x128, L.TILE_PP256x128, L.TILE_Q128x128, L.TILE_S256x128, L.TILE_S192x160, L.TILE_S256x160])
def test_gemm_temporal_conv_residual(tile):
    Bn, F_, Pp, Cc = 2, 5, 24, 64
    M = Bn * F_ * Pp
    wt = torch.randn(Cc, Cc, 3, 1, 1, generator=g(2)) * (3 * Cc) ** -0.5
    c = Case(x=rnd((M, Cc), 1), w=P.pack_tconv(wt, "cpu"), b=torch.randn(Cc, generator=g(3)), res=rnd((M, Cc), 5),
             out=torch.zeros(M, Cc, dtype=BF))

    def build(t):
        return ops.gemm_params(M, Cc, ops.temporal_segs(t["x"], Cc, Cc), t["w"], t["out"], Cc, bias=t["b"],
                               geom=ops.Geom(F=F_, P=Pp), residual=t["res"], ldr=Cc, tile=tile)
    cpu, dev = run_gemm(build, c)
    check(dev["out"], cpu["out"])
    x5 = cpu["x"].float().view(Bn, F_, Pp, Cc).permute(0, 3, 1, 2)[..., None]      # b c f p 1
    ref = torch.nn.functional.conv3d(x5, wt.to(BF).float(), cpu["b"], padding=(1, 0, 0))
    ref = ref[..., 0].permute(0, 2, 3, 1).reshape(M, Cc) + cpu["res"].float()
    check(dev["out"], ref)


@pytest.mark.parametrize("tile", [0, L.TILE_256x128, L.TILE_G128x128, L.TILE_P256x128, L.TILE_PP256x128, L.TILE_Q128x128, L.TILE_S256x128, L.TILE_S192x160, L.TILE_S256x160])
@pytest.mark.parametrize("ks", [2, 5])
def test_gemm_splitk(ks, tile):
    M, N, K = 200, 256, 1280
    c = Case(a=rnd((M, K), 1), w=rnd((N, K), 2, K ** -0.5), b=torch.randn(N, generator=g(3)), res=rnd((M, N), 5),
             out=torch.zeros(M, N, dtype=BF), ws=torch.zeros(ks * M * N))

    def build(t):
        return ops.gemm_params(M, N, ops.linear_segs([(t["a"], K, K)]), t["w"], t["out"], N, bias=t["b"],
                               residual=t["res"], ldr=N, ksplit=ks, workspace=t["ws"], tile=tile)
    cpu, dev = run_gemm(build, c)
    check(dev["out"], cpu["out"])


@pytest.mark.parametrize("tile,M,N,K,ks", [(L.TILE_X256x320, 1920, 1280, 11520, 8), (L.TILE_X256x320, 300, 640, 1280, 3), (L.TILE_X256x256, 960, 1280, 3840, 6),
                                          (L.TILE_X256x256, 257, 256, 1152, 2), (L.TILE_X256x128, 513, 128, 2304, 4), (L.TILE_X256x320, 15360, 320, 2880, 2)])
def test_gemm_splitk_wide_tile(tile, M, N, K, ks):
    """Round 4: split-K on the wide-tile kernel (gemm_xglds.hip SK): every split an even number (>= 4) of 32-deep chunks starting in the
    middle of the K walk, raw fp32 slabs, the shared reduce + epilogue pass (bias, residual) — linear rows and a 3 x 3 gather
    (K = 9 C: the splits begin inside segments), M / N tails."""
    ws = torch.zeros(ks * M * N)
    if K % 9 == 0 and (K // 9) % 8 == 0 and M % 48 == 0:            # 3 x 3 convolution over [n, h, w] = M rows
        Cc = K // 9
        n_img, hh = 3, 16
        ww = M // (n_img * hh)
        wt = torch.randn(N, Cc, 3, 3, generator=g(2)) * (9 * Cc) ** -0.5
        c = Case(x=rnd((M, Cc), 1), w=P.pack_conv3x3(wt, "cpu"), b=torch.randn(N, generator=g(3)), res=rnd((M, N), 5), out=torch.zeros(M, N, dtype=BF), ws=ws)

        def build(t):
            return ops.gemm_params(M, N, ops.conv3x3_segs([(t["x"], Cc, Cc)]), t["w"], t["out"], N, bias=t["b"], residual=t["res"], ldr=N,
                                   geom=ops.Geom(OH=hh, OW=ww, IH=hh, IW=ww), ksplit=ks, workspace=t["ws"], tile=tile)
        cpu, dev = run_gemm(build, c, cpu_ref=False)
        x4 = cpu["x"].float().view(n_img, hh, ww, Cc).permute(0, 3, 1, 2)
        ref = torch.nn.functional.conv2d(x4, wt.to(BF).float(), cpu["b"], padding=1).permute(0, 2, 3, 1).reshape(M, N) + cpu["res"].float()
        check(dev["out"], ref)
        return
    c = Case(a=rnd((M, K), 1), w=rnd((N, K), 2, K ** -0.5), b=torch.randn(N, generator=g(3)), res=rnd((M, N), 5), out=torch.zeros(M, N, dtype=BF), ws=ws)

    def build(t):
        return ops.gemm_params(M, N, ops.linear_segs([(t["a"], K, K)]), t["w"], t["out"], N, bias=t["b"], residual=t["res"], ldr=N,
                               ksplit=ks, workspace=t["ws"], tile=tile)
    cpu, dev = run_gemm(build, c, cpu_ref=False)
    check(dev["out"], cpu["a"].float() @ cpu["w"].float().t() + cpu["b"] + cpu["res"].float())


@pytest.mark.parametrize("n,H,W,Cin,N,fp32,ldo", [(2, 20, 37, 128, 4, True, 4), (3, 9, 16, 320, 4, False, 8), (2, 8, 8, 512, 8, True, 8),
                                                   (1, 33, 50, 192, 4, True, 4), (2, 16, 17, 160, 4, True, 4), (1, 5, 3, 32, 8, False, 8),
                                                   (24, 40, 64, 320, 4, True, 4)])
def test_conv_halo_few_output_channels(n, H, W, Cin, N, fp32, ldo):
    """The halo-resident 3 x 3 convolution for N <= 8 output channels (conv_halo.hip; the VAE's conv_out 128 -> 3, the encoder's
    512 -> 8, the UNet's eps head 320 -> 4): ragged tiles at the right / bottom edge, several images, channel chunks of 128 + a 64- or
    32-channel tail, fp32 and 16-bit outputs (vmv_gemm takes N % 4 == 0: 3 channels are a zero-padded fourth row) — against F.conv2d and
    the tile kernels; the dispatcher picks it."""
    M = n * H * W
    wt = torch.randn(N, Cin, 3, 3, generator=g(2)) * (9 * Cin) ** -0.5
    wp = torch.zeros((N + 3) // 4 * 4, 9 * Cin)
    wp[:N] = wt.permute(0, 2, 3, 1).reshape(N, -1)
    c = Case(x=rnd((M, Cin), 1), w=wp.to(BF), b=torch.cat([torch.randn(N, generator=g(3)), torch.zeros(8)])[: (N + 3) // 4 * 4],
             out=torch.full((M, ldo), 7.0, dtype=torch.float32 if fp32 else BF))

    def build(t, tile=L.TILE_AUTO):
        return ops.gemm_params(M, N, ops.conv3x3_segs([(t["x"], Cin, Cin)]), t["w"], t["out"], ldo, bias=t["b"], out_fp32=fp32,
                               geom=ops.Geom(OH=H, OW=W, IH=H, IW=W), tile=tile)
    import ctypes as C
    dev = c.on("cuda")
    S = ops.Stream(record=False)
    assert S.lib.vmv_gemm_pick_tile(C.byref(build(dev))) == L.TILE_HALO
    S.gemm(build(dev), "halo")
    torch.cuda.synchronize()
    x = c.t["x"].float().view(n, H, W, Cin).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(x, wt.to(BF).float(), c.t["b"][:N], padding=1).permute(0, 2, 3, 1).reshape(M, N)
    got = dev["out"].float().cpu()
    assert float((got[:, :N] - ref).abs().max()) <= (2e-3 if fp32 else 2e-2) * float(ref.abs().max()), float((got[:, :N] - ref).abs().max())
    if ldo > (N + 3) // 4 * 4:
        assert bool((got[:, (N + 3) // 4 * 4:] == 7.0).all())              # columns beyond the (4-padded) outputs are not touched
    old = c.on("cuda")                                                        # the tile kernels on the same argument block
    S.gemm(build(old, tile=L.TILE_128x64), "tile")
    torch.cuda.synchronize()
    assert float((old["out"].float()[:, :N] - dev["out"].float()[:, :N]).abs().max().cpu()) <= (2e-3 if fp32 else 2e-2) * float(ref.abs().max())


@pytest.mark.parametrize("tile", [L.TILE_S256x128, L.TILE_S192x160, L.TILE_S256x160, L.TILE_256x160, L.TILE_X256x320, L.TILE_X256x256, L.TILE_X256x128,
                                  L.TILE_X512x128])
def test_gemm_conv3x3_many_tiles(tile):
    """>= 2 tiles per persistent block (M = 2*24*40*64/2 rows), residual + per-image row vector, two sources + 1x1 skip."""
    n, IH, IW, C0, C1, N = 24, 40, 64, 64, 32, 320
    Cin = C0 + C1
    M = n * IH * IW
    wt = torch.randn(N, Cin, 3, 3, generator=g(2)) * (9 * Cin) ** -0.5
    ws = torch.randn(N, Cin, generator=g(7)) * Cin ** -0.5
    wp = torch.cat([wt.permute(0, 2, 3, 1).reshape(N, -1), ws], dim=1)
    c = Case(x0=rnd((M, C0), 1), x1=rnd((M, C1), 4), w=wp.to(BF), b=torch.randn(N, generator=g(3)),
             rv=torch.randn(n, N, generator=g(8)), out=torch.zeros(M, N, dtype=BF))

    def build(t):
        srcs = [(t["x0"], C0, C0), (t["x1"], C1, C1)]
        return ops.gemm_params(M, N, ops.conv3x3_segs(srcs) + ops.linear_segs(srcs), t["w"], t["out"], N, bias=t["b"],
                               rowvec=t["rv"], rowvec_div=IH * IW, rowvec_ld=N,
                               geom=ops.Geom(OH=IH, OW=IW, IH=IH, IW=IW), tile=tile)
    cpu, dev = run_gemm(build, c, cpu_ref=False)
    x = torch.cat([cpu["x0"].float(), cpu["x1"].float()], dim=1)
    img = x.view(n, IH, IW, Cin).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(img, wt.to(BF).float(), cpu["b"], padding=1)
    ref = ref + torch.nn.functional.conv2d(img, ws.to(BF).float()[:, :, None, None]) + cpu["rv"][:, :, None, None]
    check(dev["out"], ref.permute(0, 2, 3, 1).reshape(M, N))


@pytest.mark.parametrize("tile", [L.TILE_S256x128, L.TILE_S192x160, L.TILE_S256x160, L.TILE_X256x320, L.TILE_X256x256, L.TILE_X512x128])
def test_gemm_temporal_conv_many_tiles(tile):
    Bn, F_, Pp, Cc = 2, 24, 1280, 320
    M = Bn * F_ * Pp
    wt = torch.randn(Cc, Cc, 3, 1, 1, generator=g(2)) * (3 * Cc) ** -0.5
    c = Case(x=rnd((M, Cc), 1), w=P.pack_tconv(wt, "cpu"), b=torch.randn(Cc, generator=g(3)), res=rnd((M, Cc), 5),
             out=torch.zeros(M, Cc, dtype=BF))

    def build(t):
        return ops.gemm_params(M, Cc, ops.temporal_segs(t["x"], Cc, Cc), t["w"], t["out"], Cc, bias=t["b"],
                               geom=ops.Geom(F=F_, P=Pp), residual=t["res"], ldr=Cc, tile=tile)
    cpu, dev = run_gemm(build, c, cpu_ref=False)
    x5 = cpu["x"].float().view(Bn, F_, Pp, Cc).permute(0, 3, 1, 2)[..., None]
    ref = torch.nn.functional.conv3d(x5, wt.to(BF).float(), cpu["b"], padding=(1, 0, 0))
    ref = ref[..., 0].permute(0, 2, 3, 1).reshape(M, Cc) + cpu["res"].float()
    check(dev["out"], ref)


def _tconv_ref(x, wt, bias, Bn, F_, Pp, Cc, N):
    x5 = x.float().view(Bn, F_, Pp, Cc).permute(0, 3, 1, 2)[..., None]      # b c f p 1
    ref = torch.nn.functional.conv3d(x5, wt.to(BF).float(), bias, padding=(1, 0, 0))
    return ref[..., 0].permute(0, 2, 3, 1).reshape(Bn * F_ * Pp, N)


@pytest.mark.parametrize("Bn,F_,Pp,Cc,N,res", [(2, 24, 40, 320, 320, True), (1, 24, 21, 128, 640, False), (2, 12, 40, 64, 320, True),
                                                (1, 16, 30, 192, 320, False), (1, 20, 23, 128, 320, True), (3, 24, 8, 640, 640, False)])
def test_gemm_tfr_plain(Bn, F_, Pp, Cc, N, res):
    """Frame-resident temporal convolution (csrc/gemm_tfr.hip, VMV_TILE_TFR) WITHOUT the norm fold: a block owns all F frames of
    192 // F pixels, the three taps are row-shifted views of one LDS tile.  F = 24 / 20 / 16 / 12 (8 / 9 / 12 / 16 pixels per tile;
    F = 20 leaves 12 rows of the tile unused), pixel counts that are not a multiple of the tile (ragged last tile), one to four
    A stages of 64 channels... ten at C = 640, two column tiles at N = 640, several samples, residual — against Conv3d and the interpreter."""
    M = Bn * F_ * Pp
    wt = torch.randn(N, Cc, 3, 1, 1, generator=g(2)) * (3 * Cc) ** -0.5
    c = Case(x=rnd((M, Cc), 1), w=P.pack_tconv(wt, "cpu"), b=torch.randn(N, generator=g(3)), res=rnd((M, N), 5), out=torch.zeros(M, N, dtype=BF))

    def build(t):
        return ops.gemm_params(M, N, ops.temporal_segs(t["x"], Cc, Cc), t["w"], t["out"], N, bias=t["b"], geom=ops.Geom(F=F_, P=Pp),
                               residual=t["res"] if res else None, ldr=N if res else 0, tile=L.TILE_TFR)
    cpu, dev = run_gemm(build, c, cpu_ref=M * N * Cc < 4e8)
    ref = _tconv_ref(cpu["x"], wt, cpu["b"], Bn, F_, Pp, Cc, N) + (cpu["res"].float() if res else 0)
    check(dev["out"], ref)
    if M * N * Cc < 4e8:
        check(dev["out"], cpu["out"])
    # the same launch on the tile kernels agrees (other accumulation order only)
    dev2 = c.on("cuda")
    q = build(dev2)
    q.tile = L.TILE_AUTO if not ops.Stream(record=False).lib.vmv_gemm_tfr_ok(C.byref(q)) else L.TILE_G128x128
    ops.Stream(record=False).gemm(q, "tile")
    torch.cuda.synchronize()
    check(dev["out"], dev2["out"].cpu(), tol_l2=2e-3, tol_max=6e-3)


@pytest.mark.parametrize("Bn,F_,Pp,Cc,N,silu,res", [(2, 24, 40, 320, 320, True, True), (1, 24, 21, 128, 640, True, False), (2, 12, 19, 64, 320, False, True),
                                                     (1, 20, 23, 128, 320, True, False), (2, 24, 16, 640, 640, True, True)])
def test_gemm_tfr_groupnorm_fold(Bn, F_, Pp, Cc, N, silu, res):
    """GroupNorm over all frames -> SiLU -> Conv3d (3,1,1) (TemporalConvBlock_v2, util.py:1357-1392) as statistics + table + ONE GEMM:
    the frame-resident kernel applies elem(silu(x * scale + shift)) to its A tile in LDS.  Against (a) the two-kernel form — statistics
    + vmv_groupnorm_apply(silu) + the same kernel without the fold — which multiplies the very same rounded values (same kernel, same
    accumulation order: bitwise), (b) the interpreter, (c) F.group_norm + F.silu + Conv3d in fp32.  The zero padding of frames -1 / F
    must stay zero AFTER the norm (silu(shift) != 0), and rows of a ragged last tile must not leak into valid pixels."""
    M = Bn * F_ * Pp
    rps = F_ * Pp
    wt = torch.randn(N, Cc, 3, 1, 1, generator=g(2)) * (3 * Cc) ** -0.5
    x = (torch.randn(M, Cc, generator=g(1)) * 1.5 + 0.7).to(BF)
    gamma, beta = 1 + 0.2 * torch.randn(Cc, generator=g(6)), 0.3 * torch.randn(Cc, generator=g(7))
    c = Case(x=x, w=P.pack_tconv(wt, "cpu"), b=torch.randn(N, generator=g(3)), res=rnd((M, N), 5), out=torch.zeros(M, N, dtype=BF),
             out2=torch.zeros(M, N, dtype=BF), y=torch.zeros(M, Cc, dtype=BF), tab=torch.zeros(Bn * 2 * Cc), gamma=gamma, beta=beta,
             ws=torch.zeros(ops.gn_partial_floats(M, rps, Cc)))

    def gnp(t, y, silu_flag):
        return ops.gn_params(t["x"], Cc, Cc, M, rps, t["ws"], t["gamma"], t["beta"], 1e-5, silu_flag, y, Cc)

    def build(t, fold):
        kw = dict(gn_table=t["tab"], gn_rows_per_stat=rps, gn_silu=silu) if fold else {}
        return ops.gemm_params(M, N, ops.temporal_segs(t["x"] if fold else t["y"], Cc, Cc), t["w"], t["out"] if fold else t["out2"], N, bias=t["b"],
                               geom=ops.Geom(F=F_, P=Pp), residual=t["res"] if res else None, ldr=N if res else 0, tile=L.TILE_TFR, **kw)
    dev = c.on("cuda")
    S = ops.Stream(record=False)
    S.groupnorm_stats(gnp(dev, dev["tab"], False), "stats")
    S.groupnorm_table(gnp(dev, dev["tab"], False), "table")
    S.gemm(build(dev, True), "folded")
    S.groupnorm_apply(gnp(dev, dev["y"], silu), "apply")
    S.gemm(build(dev, False), "two-kernel")
    torch.cuda.synchronize()
    assert torch.equal(dev["out"], dev["out2"])
    cpu = c.on("cpu")
    I.groupnorm_stats(gnp(cpu, cpu["tab"], False)); I.groupnorm_table(gnp(cpu, cpu["tab"], False))
    if M * N * Cc < 4e8:
        I.gemm(build(cpu, True))
        check(dev["out"], cpu["out"])
    xn = torch.nn.functional.group_norm(x.float().view(Bn, rps, Cc).permute(0, 2, 1), 32, gamma, beta, 1e-5)
    xn = (torch.nn.functional.silu(xn) if silu else xn).permute(0, 2, 1).reshape(M, Cc).to(BF)
    ref = _tconv_ref(xn, wt, cpu["b"], Bn, F_, Pp, Cc, N) + (cpu["res"].float() if res else 0)
    check(dev["out"], ref, tol_l2=6e-3, tol_max=2.5e-2)
    # refused where it cannot run: a stat group that is not the sample, SiLU without a table, F outside 12 .. 24
    bad = build(dev, True); bad.gn_rows_per_stat = rps // 2
    assert S.lib.vmv_gemm(C.byref(bad), None) == -1
    bad = build(dev, True); bad.gn_table = None; bad.gn_silu = 1
    assert S.lib.vmv_gemm(C.byref(bad), None) == -1


def test_gemm_tfr_policy():
    """vmv_gemm_tfr_ok: the frame-resident kernel is chosen where its tiles (samples x ceil(P / (192 // F)) x N / 320) fill whole rounds
    of the 256 CUs — the first UNet level at 24 x 32 x 32 (256 tiles; 512 at 24 x 32 x 64) — and not at 24 x 40 x 64 (640 tiles = 2.5
    rounds: measured slower than the 256 x 320 tile there), at the second level (320 / 128 tiles) or on tiny nets; F < 12 and
    N % 320 != 0 are unsupported."""
    lib = ops.Stream(record=False).lib
    X = 1 << 20

    def ok(Bn, F_, Pp, Cc, N):
        p = ops.gemm_params(Bn * F_ * Pp, N, ops.temporal_segs(X, Cc, Cc), X, X, N, geom=ops.Geom(F=F_, P=Pp))
        return lib.vmv_gemm_tfr_ok(C.byref(p)), lib.vmv_gemm_pick_tile(C.byref(p))
    assert ok(2, 24, 1024, 320, 320) == (1, L.TILE_TFR) and ok(2, 24, 2048, 320, 320) == (1, L.TILE_TFR)
    assert ok(2, 24, 2560, 320, 320) == (0, L.TILE_X256x320)
    assert ok(2, 24, 640, 640, 640)[0] == 0 and ok(2, 24, 256, 640, 640)[0] == 0
    assert ok(2, 4, 64, 64, 320)[0] == 0 and ok(2, 24, 2560, 320, 256)[0] == 0


@pytest.mark.parametrize("tile", [L.TILE_X256x320, L.TILE_X256x128])
def test_gemm_wide_tile_rowvec_act_residual(tile):
    """The wide-tile kernel's staged epilogue: bias + per-group row vector + SiLU + scaled residual, 16-bit output, M tail."""
    M, N, K = 1000, 320, 256
    c = Case(a=rnd((M, K), 1), w=rnd((N, K), 2, K ** -0.5), b=torch.randn(N, generator=g(3)),
             rv=torch.randn((M + 63) // 64, 512, generator=g(4)), res=rnd((M, N), 5), out=torch.zeros(M, N, dtype=BF))

    def build(t):
        return ops.gemm_params(M, N, ops.linear_segs([(t["a"], K, K)]), t["w"], t["out"], N, bias=t["b"],
                               rowvec=t["rv"].data_ptr() + 4 * 64, rowvec_div=64, rowvec_ld=512, act=L.ACT_SILU,
                               residual=t["res"], ldr=N, res_scale=0.5, tile=tile)
    cpu, dev = run_gemm(build, c)
    check(dev["out"], cpu["out"])


@pytest.mark.parametrize("tile", [0, L.TILE_P256x128, L.TILE_G128x128, L.TILE_256x128, L.TILE_128x128])
@pytest.mark.parametrize("G,R,N,K,fp32", [(3, 512, 384, 192, False), (24, 1024, 1024, 512, True), (5, 256, 512, 256, False)])
def test_gemm_grouped_weights(G, R, N, K, fp32, tile):
    """VmvGemmParams.wgroup_rows: rows [g R, (g + 1) R) multiply the g-th weight matrix (the VAE attention's batched Q K^T /
    P V) — against one GEMM per group on the same operands, and against the interpreter."""
    M = G * R
    c = Case(a=rnd((M, K), 1), w=rnd((G * N, K), 2, K ** -0.5), b=torch.randn(N, generator=g(3)),
             out=torch.zeros(M, N) if fp32 else torch.zeros(M, N, dtype=BF))

    def build(t):
        return ops.gemm_params(M, N, ops.linear_segs([(t["a"], K, K)]), t["w"], t["out"], N, bias=t["b"], out_fp32=fp32,
                               wgroup_rows=R, wgroup_stride=N * K, tile=tile)
    cpu, dev = run_gemm(build, c)
    tol = dict(tol_l2=8e-3, tol_max=1.6e-2) if fp32 else {}       # (x TS = 0.125 for fp16 operands: 1e-3 / 2e-3)
    check(dev["out"], cpu["out"], **tol)
    ref = torch.cat([cpu["a"][gi * R:(gi + 1) * R].float() @ cpu["w"][gi * N:(gi + 1) * N].float().t() for gi in range(G)]) + cpu["b"]
    check(dev["out"], ref, **tol)


def test_gemm_grouped_weights_validation():
    import ctypes as C
    lib = L.load()
    a = rnd((512, 64), 1).cuda(); w = rnd((2 * 64, 64), 2).cuda(); o = torch.zeros(512, 64, dtype=BF, device="cuda")
    ws = torch.zeros(2 * 512 * 64, device="cuda")
    segs = ops.linear_segs([(a, 64, 64)])
    for kw in (dict(wgroup_rows=100, wgroup_stride=64 * 64), dict(wgroup_rows=256, wgroup_stride=64 * 64 + 4),
               dict(wgroup_rows=256, wgroup_stride=64 * 64, ksplit=2, workspace=ws),
               dict(wgroup_rows=256, wgroup_stride=64 * 64, tile=L.TILE_P256x160)):
        p = ops.gemm_params(512, 64, segs, w, o, 64, **kw)
        assert lib.vmv_gemm(C.byref(p), None) == -1, kw          # VMV_EINVAL


def test_gemm_geglu_many_tiles():
    M, I2, K = 70000, 1280, 320
    w = rnd((I2, K), 2, K ** -0.5)
    b = torch.randn(I2, generator=g(3))
    c = Case(a=rnd((M, K), 1), w=P.geglu_interleave(w), b=P.geglu_interleave(b), out=torch.zeros(M, I2 // 2, dtype=BF))

    def build(t):
        return ops.gemm_params(M, I2, ops.linear_segs([(t["a"], K, K)]), t["w"], t["out"], I2 // 2, bias=t["b"],
                               epilogue=L.EPI_GEGLU, tile=L.TILE_P256x128)
    cpu, dev = run_gemm(build, c, cpu_ref=False)
    h = cpu["a"].float() @ w.float().t() + b
    x, gate = h.chunk(2, dim=-1)
    check(dev["out"], x * torch.nn.functional.gelu(gate))


@pytest.mark.parametrize("M,I2,K", [(7680, 10240, 1280), (1000, 1280, 320), (257, 512, 128)])
def test_gemm_geglu_wide_tile(M, I2, K):
    """GEGLU without a folded LayerNorm on the 256 x 256 wide tile (x | gate column tiles = adjacent accumulators of one lane)."""
    w = rnd((I2, K), 2, K ** -0.5)
    b = torch.randn(I2, generator=g(3))
    c = Case(a=rnd((M, K), 1), w=P.geglu_interleave(w), b=P.geglu_interleave(b), out=torch.zeros(M, I2 // 2, dtype=BF))

    def build(t):
        return ops.gemm_params(M, I2, ops.linear_segs([(t["a"], K, K)]), t["w"], t["out"], I2 // 2, bias=t["b"],
                               epilogue=L.EPI_GEGLU, tile=L.TILE_X256x256)
    cpu, dev = run_gemm(build, c, cpu_ref=False)
    h = cpu["a"].float() @ w.float().t() + b
    x, gate = h.chunk(2, dim=-1)
    check(dev["out"], x * torch.nn.functional.gelu(gate))


def test_gemm_rejects_bad_arguments():
    lib = L.load()
    t = Case(a=rnd((64, 64), 1), w=rnd((64, 64), 2), out=torch.zeros(64, 64, dtype=BF)).on("cuda")
    import ctypes as C
    p = ops.gemm_params(64, 64, ops.linear_segs([(t["a"], 64, 64)]), t["w"], t["out"], 64)
    p.N = 63
    assert lib.vmv_gemm(C.byref(p), None) == -1          # VMV_EINVAL
    p = ops.gemm_params(64, 64, ops.linear_segs([(t["a"].data_ptr() + 2, 64, 64)]), t["w"], t["out"], 64)
    assert lib.vmv_gemm(C.byref(p), None) == -2          # VMV_EALIGN
    p = ops.gemm_params(64, 64, ops.linear_segs([(t["a"], 64, 64)]), None, t["out"], 64)
    assert lib.vmv_gemm(C.byref(p), None) == -3          # VMV_ENULL


# ------------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("rows,rps,C0,C1,silu,eps", [(6 * 80, 80, 320, 0, True, 1e-5), (2 * 3 * 64, 3 * 64, 64, 0, False, 1e-6),
                                                     (4 * 100, 100, 1280, 640, True, 1e-5), (2 * 1000, 1000, 32, 0, True, 1e-5),
                                                     (3 * 50, 50, 2560, 0, True, 1e-5)])
def test_groupnorm(rows, rps, C0, C1, silu, eps):
    Cc = C0 + C1
    c = Case(x=rnd((rows, C0), 1) + 0.5, x1=rnd((rows, max(C1, 8)), 2, 2.0), gamma=1 + 0.1 * torch.randn(Cc, generator=g(3)),
             beta=0.1 * torch.randn(Cc, generator=g(4)), y=torch.zeros(rows, Cc, dtype=BF),
             part=torch.zeros(ops.gn_partial_floats(rows, rps, Cc) + 64))

    def build(t):
        return ops.gn_params(t["x"], C0, C0, rows, rps, t["part"], t["gamma"], t["beta"], eps, silu, t["y"], Cc,
                             x1=t["x1"] if C1 else None, ld1=max(C1, 8) if C1 else 0, C1=C1)
    cpu = c.on("cpu")
    I.groupnorm_stats(build(cpu))
    I.groupnorm(build(cpu))
    dev = c.on("cuda")
    S = ops.Stream(record=False)
    S.groupnorm_stats(build(dev))                   # the general two-launch form, explicitly
    S.groupnorm_apply(build(dev))
    torch.cuda.synchronize()
    check(dev["y"], cpu["y"], tol_l2=5e-3, tol_max=1.5e-2)
    dev2 = c.on("cuda")
    S.groupnorm(build(dev2))                        # auto: one fused launch where the stat group fits on chip
    torch.cuda.synchronize()
    check(dev2["y"], cpu["y"], tol_l2=5e-3, tol_max=1.5e-2)


@pytest.mark.parametrize("rows,rps,Cc,silu", [(2 * 61440, 61440, 320, True), (2 * 960, 960, 1280, False), (3 * 500, 500, 64, True)])
def test_groupnorm_prefolded_totals(rows, rps, Cc, silu):
    """Long stat groups (the all-frame norms): the stats blocks add their sums to two-limb 64-bit fixed-point integer accumulators
    totals[stat][32][GN_NREP][GN_REC] (order-independent => deterministic, no fold launch); apply reads them and clears the next norm's.
    Same statistics as the plain two-launch path up to summation order (the two limbs hold each block's fp32 sum exactly)."""
    x = rnd((rows, Cc), 21, 1.3).cuda() + 0.1
    gamma, beta = (1 + 0.1 * torch.randn(Cc, generator=g(3))).cuda(), (0.1 * torch.randn(Cc, generator=g(4))).cuda()
    part = torch.zeros(ops.gn_partial_floats(rows, rps, Cc) + 64, device="cuda")
    tot = torch.zeros(2, 64 * ops.GN_TOT, dtype=torch.int64, device="cuda")
    y0, y1 = torch.zeros(rows, Cc, dtype=BF, device="cuda"), torch.zeros(rows, Cc, dtype=BF, device="cuda")
    S = ops.Stream(record=False)
    p0 = ops.gn_params(x, Cc, Cc, rows, rps, part, gamma, beta, 1e-5, silu, y0, Cc)
    S.groupnorm_stats(p0); S.groupnorm_apply(p0)                          # (the plain two-launch form, explicitly)
    S.groupnorm(ops.gn_params(x, Cc, Cc, rows, rps, part, gamma, beta, 1e-5, silu, y1, Cc, totals=tot[0],
                              totals_clear=tot[1], clear_count=tot[1].numel()))
    torch.cuda.synchronize()
    check(y1, y0.float().cpu(), tol_l2=1e-4 / TS, tol_max=2e-2)             # (<= 1 ulp where the rounding flips)
    assert int(tot[0].abs().sum()) > 0 and int(tot[1].abs().sum()) == 0
    y2 = torch.zeros_like(y1)                                               # next norm: the other buffer, which apply cleared
    S.groupnorm(ops.gn_params(x, Cc, Cc, rows, rps, part, gamma, beta, 1e-5, silu, y2, Cc, totals=tot[1],
                              totals_clear=tot[0], clear_count=tot[0].numel()))
    torch.cuda.synchronize()
    assert torch.equal(y1, y2)                                           # deterministic (integer accumulation)
    assert int(tot[0].abs().sum()) == 0


@pytest.mark.parametrize("R,B,rps_loc,Cc", [(2, 2, 96, 320), (8, 2, 15, 1280), (4, 1, 640, 64)])
def test_groupnorm_sharded_statistics(R, B, rps_loc, Cc):
    """Frame-parallel 5-D norm: R ranks each hold rps_loc rows of every stat group; stats per shard, partial sums gathered
    as [R][nstat][nchunk][64], apply with fold_ranks=R.  Reference = torch group_norm over the UNsharded rows."""
    xs = [rnd((B * rps_loc, Cc), 10 + r, 1.5) + 0.2 * r for r in range(R)]
    gamma, beta = 1 + 0.1 * torch.randn(Cc, generator=g(3)), 0.1 * torch.randn(Cc, generator=g(4))
    full = torch.cat([x.view(B, rps_loc, Cc) for x in xs], dim=1).float()            # [B][R*rps_loc][C]
    ref = torch.nn.functional.group_norm(full.permute(0, 2, 1), 32, gamma, beta, 1e-5).permute(0, 2, 1)
    ref = torch.nn.functional.silu(ref)
    nfl = ops.gn_partial_floats(B * rps_loc, rps_loc, Cc)
    part = torch.zeros(nfl, device="cuda")
    S = ops.Stream(record=False)
    xd = [x.cuda() for x in xs]
    gd, bd = gamma.cuda(), beta.cuda()
    # every "rank" folds its own chunks into its records; the [R][B][32][GN_NREP][GN_REC] records are "gathered" (each relative to the
    # rank's own pilot — the shards' first elements differ — and moved to rank 0's by the apply pass)
    nrec = B * ops.GN_TOT
    tot_all = torch.zeros(R * nrec, dtype=torch.int64, device="cuda")
    ys = [torch.zeros(B * rps_loc, Cc, dtype=BF, device="cuda") for _ in range(R)]
    for r in range(R):
        S.groupnorm_stats(ops.gn_params(xd[r], Cc, Cc, B * rps_loc, rps_loc, part, gd, bd, 1e-5, True, ys[r], Cc,
                                        totals=tot_all[r * nrec:]))
    for r in range(R):
        S.groupnorm_apply(ops.gn_params(xd[r], Cc, Cc, B * rps_loc, rps_loc, part, gd, bd, 1e-5, True, ys[r], Cc,
                                        fold_ranks=R, totals=tot_all))
    torch.cuda.synchronize()
    got = torch.cat([y.view(B, rps_loc, Cc) for y in ys], dim=1)
    check(got, ref.to(BF), tol_l2=5e-3, tol_max=3e-2)
    import ctypes as C
    bad = ops.gn_params(xd[0], Cc, Cc, B * rps_loc, rps_loc, part, gd, bd, 1e-5, True, ys[0], Cc, fold_ranks=R)    # shards without records
    assert S.lib.vmv_groupnorm_apply(C.byref(bad), None) == -1


@pytest.mark.parametrize("kind", ["tiny", "offset", "huge", "mixed"])
@pytest.mark.parametrize("path", ["partial", "totals", "fused"])
def test_groupnorm_statistics_are_shift_and_scale_safe(kind, path):
    """the statistics must survive activations that are tiny (1e-3 N(0,1): round 2's single 2^-12 fixed-point limb lost
    the variance), far from zero (100 + N(0,1): E[x^2] - mean^2 cancels in fp32) or huge (3e3 N(0,1); 1e6-sized in the bf16
    build, whose range allows it) — on all three forms (per-chunk partial sums, integer totals, the one-launch LDS form).
    Checked against fp64 group statistics of the SAME 16-bit inputs: normalised output within 2e-3 (storage rounding included)."""
    rows, rps, Cc = 2 * 4800, 4800, 320
    base = torch.randn(rows, Cc, generator=g(31))
    if kind == "tiny":
        x = 1e-3 * base
    elif kind == "offset":
        x = 100.0 + base
    elif kind == "huge":
        x = (1e6 if BF == torch.bfloat16 else 3e3) * base
    else:       # groups of very different scale side by side in one tensor (per-group pilots / exponents)
        x = base * torch.logspace(-3, 3, Cc)[None, :]
    x = x.to(BF)
    if path == "fused":
        rows, rps = 2 * 160, 160
        x = x[:rows].contiguous()
    xd = x.cuda()
    gamma, beta = torch.ones(Cc), torch.zeros(Cc)
    y = torch.zeros(rows, Cc, dtype=BF, device="cuda")
    part = torch.zeros(ops.gn_partial_floats(rows, rps, Cc) + 64, device="cuda")
    tot = torch.zeros(2, 64 * ops.GN_TOT, dtype=torch.int64, device="cuda")
    S = ops.Stream(record=False)
    pr = ops.gn_params(xd, Cc, Cc, rows, rps, part, gamma.cuda(), beta.cuda(), 1e-12, False, y, Cc,
                       **(dict(totals=tot[0], totals_clear=tot[1], clear_count=tot[1].numel()) if path == "totals" else {}))
    if path == "fused":
        S.groupnorm_fused(pr, ops.gn_fused_cols(rps, Cc))
    else:
        S.groupnorm_stats(pr); S.groupnorm_apply(pr)
    torch.cuda.synchronize()
    xg = x.double().view(rows // rps, rps, 32, Cc // 32)
    mean = xg.mean(dim=(1, 3), keepdim=True)
    var = ((xg - mean) ** 2).mean(dim=(1, 3), keepdim=True)
    ref = ((xg - mean) / torch.sqrt(var + 1e-12)).view(rows, Cc)
    got = y.double().cpu()
    assert torch.isfinite(got).all()
    err = float((got - ref).abs().max())
    assert err < (2e-3 if BF == torch.float16 else 1.6e-2) * float(ref.abs().max()), (kind, path, err)


def test_permute_copy_matches_torch():
    """The four pack / unpack permutations of the frame-major <-> pixel-major layout switch (unet_engine._switch)."""
    R, B, Fl, Pl, Cc = 4, 2, 3, 10, 64
    cv = Cc // 8
    x = rnd((B * Fl * R * Pl, Cc), 7).cuda()
    S = ops.Stream(record=False)
    out = torch.zeros_like(x)
    S.copy(ops.copy_params(x, out, R, B * Fl, 1, Pl * cv, Pl * cv, R * Pl * cv))      # send[s][bf][p c] <- x[bf][s][p c]
    torch.cuda.synchronize()
    assert torch.equal(out.view(R, B * Fl, Pl, Cc), x.view(B * Fl, R, Pl, Cc).permute(1, 0, 2, 3))
    blk = Fl * Pl * cv
    S.copy(ops.copy_params(x, out, B, R, 1, blk, blk, B * blk))                       # y[b][r][...] <- recv[r][b][...]
    torch.cuda.synchronize()
    assert torch.equal(out.view(B, R, Fl * Pl, Cc), x.view(R, B, Fl * Pl, Cc).permute(1, 0, 2, 3))
    S.copy(ops.copy_params(x, out, R, B, 1, blk, blk, R * blk))                       # send[r][b][...] <- x[b][r][...]
    torch.cuda.synchronize()
    assert torch.equal(out.view(R, B, Fl * Pl, Cc), x.view(B, R, Fl * Pl, Cc).permute(1, 0, 2, 3))
    S.copy(ops.copy_params(x, out, B * Fl, R, 1, Pl * cv, Pl * cv, B * Fl * Pl * cv))  # y[bf][s][p c] <- recv[s][bf][p c]
    torch.cuda.synchronize()
    assert torch.equal(out.view(B * Fl, R, Pl, Cc), x.view(R, B * Fl, Pl, Cc).permute(1, 0, 2, 3))
    bad = ops.copy_params(x, out, 0, 1, 1, 1, 0, 0)
    import ctypes
    assert L.load().vmv_permute_copy(ctypes.byref(bad), None) == -1          # VMV_EINVAL

@pytest.mark.parametrize("rows,rps,C0,C1,silu", [(2 * 3 * 160, 160, 1280, 0, True), (2 * 40, 40, 1280, 1280, True),
                                                  (2 * 960, 960, 1280, 0, False), (4 * 640, 640, 640, 0, True),
                                                  (3 * 160, 160, 1280, 640, True)])
def test_groupnorm_fused(rows, rps, C0, C1, silu):
    """vmv_groupnorm_fused (one launch: LDS-resident stat group, two-pass statistics) vs the interpreter, incl. the
    decoder's two-source concat, the all-frame groups of L3 and a large DC offset (|mean| = 30 std: the single-pass
    E[x^2] - mean^2 of the general path would lose ~3 digits there; the two-pass form must not)."""
    C = C0 + C1
    cols = ops.gn_fused_cols(rps, C)
    assert cols > 0 and cols % (C // 32) == 0
    for offset in (0.0, 30.0):
        def build(t):
            return ops.gn_params(t["x"], C0, C0, rows, rps, t["x"], t["gamma"], t["beta"], 1e-5, silu, t["y"], C,
                                 x1=t["x1"] if C1 else None, ld1=C1, C1=C1)
        case = Case(x=(rnd((rows, C0), 1, dtype=torch.float32) + offset).to(BF),
                    x1=(rnd((rows, max(C1, 8)), 2, dtype=torch.float32) * 2.0 - offset).to(BF),
                    gamma=rnd((C,), 3, dtype=torch.float32) + 1.0, beta=rnd((C,), 4, dtype=torch.float32),
                    y=torch.zeros(rows, C, dtype=BF))
        cpu = case.on("cpu")
        I.groupnorm_fused(build(cpu))
        dev = case.on("cuda")
        S = ops.Stream(record=False)
        S.groupnorm_fused(build(dev), cols, "t")
        torch.cuda.synchronize()
        check(dev["y"], cpu["y"], tol_l2=5e-3, tol_max=2e-2)
        # the auto-selecting entry takes the same path
        dev2 = case.on("cuda")
        S.groupnorm(build(dev2), "t")
        torch.cuda.synchronize()
        assert torch.equal(dev2["y"], dev["y"])
    assert ops.gn_fused_cols(2560, 320) == 0 and ops.gn_fused_cols(3840, 1280) == 0      # too large: two-kernel form



@pytest.mark.parametrize("rows,Cc", [(37, 320), (1000, 1280), (5, 512), (64, 64), (3, 2048)])
def test_layernorm(rows, Cc):
    c = Case(x=rnd((rows, Cc), 1, 2.0) + 0.3, gamma=1 + 0.1 * torch.randn(Cc, generator=g(3)),
             beta=0.1 * torch.randn(Cc, generator=g(4)), y=torch.zeros(rows, Cc, dtype=BF))

    def build(t):
        return ops.ln_params(t["x"], Cc, t["y"], Cc, t["gamma"], t["beta"], rows, Cc)
    cpu = c.on("cpu")
    I.layernorm(build(cpu))
    dev = c.on("cuda")
    ops.Stream(record=False).layernorm(build(dev))
    torch.cuda.synchronize()
    check(dev["y"], cpu["y"], tol_l2=5e-3, tol_max=1.5e-2)


@pytest.mark.parametrize("rows,Cc", [(300, 320), (1000, 640), (77, 1280)])
def test_layernorm_stats_out(rows, Cc):
    """(mean, rstd) per row — the statistics pass of a LayerNorm folded into its consumer GEMM."""
    x = (rnd((rows, Cc), 1, 2.0).float() + 0.3).to(BF).cuda()
    st = torch.zeros(rows, 2, device="cuda")
    ops.Stream(record=False).layernorm(ops.ln_params(x, Cc, None, 0, None, None, rows, Cc, 1e-5, stats_out=st))
    torch.cuda.synchronize()
    xf = x.float()
    ref = torch.stack([xf.mean(dim=1), torch.rsqrt(xf.var(dim=1, unbiased=False) + 1e-5)], dim=1)
    assert torch.allclose(st, ref, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("M,N,K,geglu,tile", [(300, 960, 320, False, 0), (1000, 640, 640, False, L.TILE_P256x160),
                                               (70000, 320, 320, False, 0), (513, 1280, 320, True, 0),
                                               (70000, 1280, 320, True, L.TILE_P256x128), (257, 128, 192, False, L.TILE_128x128),
                                               (300, 512, 64, True, L.TILE_128x128), (2000, 3840, 1280, False, 0),
                                               # the A-stationary deferred-epilogue kernel: M tail, several panels per
                                               # block (M > 128 x 256 CUs), N tail (N = 1000: not a multiple of 160 / 128)
                                               (70001, 960, 320, False, L.TILE_A128x160), (70001, 2560, 320, True, L.TILE_A128x128),
                                               (1000, 1000, 320, False, L.TILE_A128x160), (300, 640, 256, True, L.TILE_A128x128),
                                               (129, 320, 320, False, L.TILE_A128x128), (50000, 960, 320, False, 0),
                                               (50000, 2560, 320, True, 0),
                                               # round 4: the wide tile's 256 x 256 form with the folded LayerNorm and / or GEGLU in its epilogue
                                               # (gemm_xglds.hip EPI): M / N tails, several tiles per block column, K = 128 .. 1280, the L2 shapes
                                               (7680, 3840, 1280, False, L.TILE_X256x256), (7680, 10240, 1280, True, L.TILE_X256x256),
                                               (300, 512, 128, False, L.TILE_X256x256), (513, 768, 192, True, L.TILE_X256x256),
                                               (1000, 1000, 320, False, L.TILE_X256x256), (2001, 1280, 640, True, L.TILE_X256x256),
                                               (7680, 3840, 1280, False, 0), (7680, 10240, 1280, True, 0),
                                               # round 6: the same epilogues in 256-thread blocks, two per CU (gemm_xglds.hip WNV = 4)
                                               (7680, 3840, 1280, False, L.TILE_Y256x128), (7680, 10240, 1280, True, L.TILE_Y256x128),
                                               (300, 512, 128, False, L.TILE_Y256x128), (513, 768, 192, True, L.TILE_Y256x128),
                                               (1000, 1000, 320, False, L.TILE_Y256x128), (2001, 1280, 640, True, L.TILE_Y256x128),
                                               # round 6: the wide-wave register-staged kernel (gemm_wreg.hip)
                                               (7680, 3840, 1280, False, L.TILE_W256x256), (7680, 10240, 1280, True, L.TILE_W256x256), (513, 768, 192, True, L.TILE_W256x256),
                                               (1000, 1000, 320, False, L.TILE_W256x256)])
def test_gemm_layernorm_folded(M, N, K, geglu, tile):
    """y = Linear(LayerNorm(x)) as ONE GEMM on the raw rows (packing.fold_layernorm + rowstat / colsum epilogue) against
    the unfused definition, x with a large per-row offset (mean / sigma ~ 3) to exercise the cancellation."""
    x = (rnd((M, K), 1, 1.5).float() + 4.0 * torch.randn(M, 1, generator=g(9))).to(BF)
    w = torch.randn(N, K, generator=g(2)) * K ** -0.5
    b = torch.randn(N, generator=g(3))
    gamma, beta = 1 + 0.2 * torch.randn(K, generator=g(4)), 0.2 * torch.randn(K, generator=g(5))
    wf, bf, cs = P.fold_layernorm(w, b, gamma, beta)
    if geglu:
        wf, bf, cs = P.geglu_interleave(wf), P.geglu_interleave(bf), P.geglu_interleave(cs)
    No = N // 2 if geglu else N
    xd = x.cuda()
    st = torch.zeros(M, 2, device="cuda")
    out = torch.zeros(M, No, dtype=BF, device="cuda")
    S = ops.Stream(record=False)
    S.layernorm(ops.ln_params(xd, K, None, 0, None, None, M, K, 1e-5, stats_out=st))
    S.gemm(ops.gemm_params(M, N, ops.linear_segs([(xd, K, K)]), wf.cuda(), out, No, bias=bf.cuda(), rowstat=st, colsum=cs.cuda(),
                           epilogue=L.EPI_GEGLU if geglu else L.EPI_NONE, tile=tile))
    torch.cuda.synchronize()
    ln = torch.nn.functional.layer_norm(x.float(), (K,), gamma, beta, 1e-5)
    h = ln @ w.t() + b
    if geglu:
        a, gate = h.chunk(2, dim=-1)
        h = a * torch.nn.functional.gelu(gate)
    check(out, h, tol_l2=6e-3, tol_max=2e-2)

@pytest.mark.parametrize("M,N,K,geglu,tile", [(300, 960, 320, False, 0), (1000, 640, 640, False, L.TILE_P256x160),
                                               (70000, 960, 320, False, 0), (513, 1280, 320, True, 0),
                                               (70000, 2560, 320, True, L.TILE_P256x128), (257, 128, 192, False, 0),
                                               (300, 512, 72, True, 0), (2000, 3840, 1280, False, 0), (61, 1920, 640, False, 0)])
def test_gemm_layernorm_inline(M, N, K, geglu, tile):
    """The same folded LayerNorm with the row statistics accumulated inside the GEMM's main loop (VmvGemmParams.ln_eps: no
    statistics pass) against the unfused definition — rows with a large offset (mean / sigma ~ 3), K tails (72, 192: chunks
    of 64 with zero-filled lanes), M tails, several tiles per block; and bitwise equal over two runs."""
    import ctypes as C
    x = (rnd((M, K), 1, 1.5).float() + 4.0 * torch.randn(M, 1, generator=g(9))).to(BF)
    w = torch.randn(N, K, generator=g(2)) * K ** -0.5
    b = torch.randn(N, generator=g(3))
    gamma, beta = 1 + 0.2 * torch.randn(K, generator=g(4)), 0.2 * torch.randn(K, generator=g(5))
    wf, bf, cs = P.fold_layernorm(w, b, gamma, beta)
    if geglu:
        wf, bf, cs = P.geglu_interleave(wf), P.geglu_interleave(bf), P.geglu_interleave(cs)
    No = N // 2 if geglu else N
    xd, wd, bd, cd = x.cuda(), wf.cuda(), bf.cuda(), cs.cuda()
    S = ops.Stream(record=False)
    outs = []
    for _ in range(2):
        out = torch.zeros(M, No, dtype=BF, device="cuda")
        p = ops.gemm_params(M, N, ops.linear_segs([(xd, K, K)]), wd, out, No, bias=bd, colsum=cd, ln_eps=1e-5,
                            epilogue=L.EPI_GEGLU if geglu else L.EPI_NONE, tile=tile)
        assert S.lib.vmv_gemm_ln_inline_ok(C.byref(p)) == 1
        S.gemm(p)
        torch.cuda.synchronize()
        outs.append(out)
    ln = torch.nn.functional.layer_norm(x.float(), (K,), gamma, beta, 1e-5)
    h = ln @ w.t() + b
    if geglu:
        a, gate = h.chunk(2, dim=-1)
        h = a * torch.nn.functional.gelu(gate)
    check(outs[0], h, tol_l2=6e-3, tol_max=2e-2)
    assert torch.equal(outs[0], outs[1])


def test_gemm_layernorm_inline_eligibility():
    """vmv_gemm_ln_inline_ok / vmv_gemm agree on what the in-loop statistics cannot serve: residual, split-K, fp32 output,
    several segments, a forced non-persistent tile."""
    import ctypes as C
    lib = L.load()
    x = rnd((256, 128), 1).cuda(); w = rnd((128, 128), 2).cuda(); o = torch.zeros(256, 128, dtype=BF, device="cuda")
    cs = torch.zeros(128, device="cuda"); ws = torch.zeros(2 * 256 * 128, device="cuda")
    segs = ops.linear_segs([(x, 128, 128)])
    ok = ops.gemm_params(256, 128, segs, w, o, 128, colsum=cs, ln_eps=1e-5)
    assert lib.vmv_gemm_ln_inline_ok(C.byref(ok)) == 1
    bad = [ops.gemm_params(256, 128, segs, w, o, 128, colsum=cs, ln_eps=1e-5, residual=o, ldr=128),
           ops.gemm_params(256, 128, segs, w, o, 128, colsum=cs, ln_eps=1e-5, ksplit=2, workspace=ws),
           ops.gemm_params(256, 128, segs, w, ws, 128, colsum=cs, ln_eps=1e-5, out_fp32=True),
           ops.gemm_params(256, 128, ops.linear_segs([(x, 128, 64), (x, 128, 64)]), w, o, 128, colsum=cs, ln_eps=1e-5),
           ops.gemm_params(256, 128, segs, w, o, 128, colsum=cs, ln_eps=1e-5, tile=L.TILE_128x128)]
    for p in bad:
        assert lib.vmv_gemm_ln_inline_ok(C.byref(p)) == 0
        assert lib.vmv_gemm(C.byref(p), None) == -1          # VMV_EINVAL


@pytest.mark.parametrize("M,N,K,geglu,bias,tile", [(40000, 640, 320, False, True, L.TILE_A128x160), (513, 1280, 320, True, True, L.TILE_A128x128),
                                                   (900, 960, 256, False, False, L.TILE_A128x160), (33000, 1280, 320, True, False, L.TILE_A128x128)])
def test_gemm_astat_plain(M, N, K, geglu, bias, tile):
    """gemm_astat.hip without the LayerNorm fold (bias on / off, GEGLU on / off) against the interpreter; two runs bitwise
    identical (fixed accumulation order, no atomics)."""
    c = Case(a=rnd((M, K), 1), w=rnd((N, K), 2, K ** -0.5), b=torch.randn(N, generator=g(3)),
             out=torch.zeros(M, N // 2 if geglu else N, dtype=BF))

    def build(t):
        return ops.gemm_params(M, N, ops.linear_segs([(t["a"], K, K)]), t["w"], t["out"], t["out"].shape[1],
                               bias=t["b"] if bias else None, epilogue=L.EPI_GEGLU if geglu else L.EPI_NONE, tile=tile)
    cpu = c.on("cpu")
    I.gemm(build(cpu))
    dev = c.on("cuda")
    S = ops.Stream(record=False)
    S.gemm(build(dev), "t")
    torch.cuda.synchronize()
    check(dev["out"], cpu["out"])
    dev2 = c.on("cuda")
    S.gemm(build(dev2), "t")
    torch.cuda.synchronize()
    assert torch.equal(dev2["out"], dev["out"])



# ------------------------------------------------------------------------------------------------- row-stationary GEMM
RS_LIN = [(1000, 960, 320, L.TILE_RS512), (300, 320, 320, L.TILE_RS256), (70000, 320, 320, L.TILE_RS512), (513, 64, 320, L.TILE_RS256),
          (1000, 1920, 640, L.TILE_RS256), (513, 640, 640, L.TILE_RS256), (40000, 640, 640, L.TILE_RS), (70001, 960, 320, L.TILE_RS),
          (2000, 2560, 320, L.TILE_RS512), (35000, 128, 640, L.TILE_RS256),
          (1000, 1536, 512, L.TILE_RS256), (61440, 512, 512, L.TILE_RS), (257, 320, 512, L.TILE_RS256)]      # K = 512: the init TemporalTransformer (round 6)


@pytest.mark.parametrize("M,N,K,tile", RS_LIN)
@pytest.mark.parametrize("bias,res", [(True, False), (False, False), (True, True)])
def test_gemm_rs_linear(M, N, K, tile, bias, res):
    """gemm_rs.hip (rows resident in registers, W streamed through the LDS ring, outputs per 32-column pair): M tails, one to
    many blocks, every column split the launcher picks for these shapes, bias on / off, residual with res_scale; two runs
    bitwise identical."""
    c = Case(a=rnd((M, K), 1), w=rnd((N, K), 2, K ** -0.5), b=torch.randn(N, generator=g(3)), res=rnd((M, N), 5),
             out=torch.zeros(M, N, dtype=BF))

    def build(t):
        kw = dict(residual=t["res"], ldr=N, res_scale=0.5) if res else {}
        return ops.gemm_params(M, N, ops.linear_segs([(t["a"], K, K)]), t["w"], t["out"], N, bias=t["b"] if bias else None, tile=tile, **kw)
    cpu = c.on("cpu")
    if M <= 2000:
        I.gemm(build(cpu))
        ref = cpu["out"]
    else:
        ref = cpu["a"].float() @ cpu["w"].float().t() + (cpu["b"] if bias else 0) + (0.5 * cpu["res"].float() if res else 0)
    dev = c.on("cuda")
    S = ops.Stream(record=False)
    S.gemm(build(dev), "t")
    torch.cuda.synchronize()
    check(dev["out"], ref)
    dev2 = c.on("cuda")
    S.gemm(build(dev2), "t")
    torch.cuda.synchronize()
    assert torch.equal(dev2["out"], dev["out"])


@pytest.mark.parametrize("M,N,K,tile", [(1000, 2560, 320, L.TILE_RS512), (300, 128, 320, L.TILE_RS256), (70000, 2560, 320, L.TILE_RS), (5000, 4096, 512, L.TILE_RS256),
                                         (700, 5120, 640, L.TILE_RS256), (33000, 1280, 640, L.TILE_RS)])
@pytest.mark.parametrize("ln", [False, True])
def test_gemm_rs_geglu_layernorm(M, N, K, tile, ln):
    """GEGLU (x | gate pairs of one body stored as one 32-column unit) with and without the LayerNorm applied to the resident
    rows (colsum + ln_eps, two-pass statistics in the kernel) against the unfused definition; rows carry a large offset
    (mean / sigma ~ 3)."""
    import ctypes as C
    x = (rnd((M, K), 1, 1.5).float() + 4.0 * torch.randn(M, 1, generator=g(9))).to(BF)
    w = torch.randn(N, K, generator=g(2)) * K ** -0.5
    b = torch.randn(N, generator=g(3))
    gamma, beta = 1 + 0.2 * torch.randn(K, generator=g(4)), 0.2 * torch.randn(K, generator=g(5))
    if ln:
        wf, bf, cs = P.fold_layernorm(w, b, gamma, beta)
    else:
        wf, bf, cs = w.to(BF).float(), b, None
    wp, bp = P.geglu_interleave(wf), P.geglu_interleave(bf)
    xd = x.cuda()
    out = torch.zeros(M, N // 2, dtype=BF, device="cuda")
    kw = dict(colsum=P.geglu_interleave(cs).cuda(), ln_eps=1e-5) if ln else {}
    p = ops.gemm_params(M, N, ops.linear_segs([(xd, K, K)]), wp.to(BF).cuda(), out, N // 2, bias=bp.cuda(), epilogue=L.EPI_GEGLU, tile=tile, **kw)
    S = ops.Stream(record=False)
    if ln:
        assert S.lib.vmv_gemm_ln_inline_ok(C.byref(p)) == 1
    S.gemm(p)
    torch.cuda.synchronize()
    xin = torch.nn.functional.layer_norm(x.float(), (K,), gamma, beta, 1e-5) if ln else x.float()
    h = xin @ (w if ln else wf).t() + b
    a, gate = h.chunk(2, dim=-1)
    check(out, a * torch.nn.functional.gelu(gate), tol_l2=6e-3, tol_max=2e-2)


@pytest.mark.parametrize("M,N,K,tile", [(1000, 960, 320, L.TILE_RS512), (70000, 960, 320, L.TILE_RS), (61, 1920, 640, L.TILE_RS256), (3000, 1536, 512, L.TILE_RS256),
                                         (31000, 1920, 640, L.TILE_RS), (300, 320, 320, L.TILE_RS256)])
def test_gemm_rs_layernorm(M, N, K, tile):
    """y = Linear(LayerNorm(x)) on the row-stationary kernel: statistics and normalisation from the resident rows (no rowstat;
    a rowstat that IS passed is ignored).  Rows with a large offset AND rows that are tiny (1e-3) or huge (3e3): the
    two-pass variance has no E[x^2] - mean^2 cancellation; zero rows give bias' exactly."""
    x = rnd((M, K), 1, 1.5).float() + 4.0 * torch.randn(M, 1, generator=g(9))
    x[1::7] *= 1e-3
    x[2::7] = x[2::7] * 30 + 3e3
    x[3] = 0
    x = x.to(BF)
    w = torch.randn(N, K, generator=g(2)) * K ** -0.5
    b = torch.randn(N, generator=g(3))
    gamma, beta = 1 + 0.2 * torch.randn(K, generator=g(4)), 0.2 * torch.randn(K, generator=g(5))
    wf, bf, cs = P.fold_layernorm(w, b, gamma, beta)
    xd = x.cuda()
    junk = torch.full((M, 2), 7.0, device="cuda")
    outs = []
    for rowstat in (None, junk):
        out = torch.zeros(M, N, dtype=BF, device="cuda")
        S = ops.Stream(record=False)
        S.gemm(ops.gemm_params(M, N, ops.linear_segs([(xd, K, K)]), wf.cuda(), out, N, bias=bf.cuda(), colsum=cs.cuda(), ln_eps=1e-5,
                               rowstat=rowstat, tile=tile))
        torch.cuda.synchronize()
        outs.append(out)
    ref = torch.nn.functional.layer_norm(x.float(), (K,), gamma, beta, 1e-5) @ w.t() + b
    check(outs[0], ref, tol_l2=6e-3, tol_max=2e-2)
    assert torch.equal(outs[0], outs[1])


def _tqa_case(nb, F_, Pp, heads, ln, seed=0):
    """Fused q | k | v projection + per-pixel temporal attention (csrc/gemm_tqa.hip, VMV_EPI_TATTN): inputs + the torch reference
    (LayerNorm -> q, k, v = Linear -> 16-bit -> softmax(q k^T / 8) v over the F frames of every (sample, pixel, head))."""
    K, inner = 320, 64 * heads
    M = nb * F_ * Pp
    x = (rnd((M, K), seed + 1, 1.2).float() + 0.7 * torch.randn(M, 1, generator=g(seed + 9))).to(BF)
    w = torch.randn(3 * inner, K, generator=g(seed + 2)) * (1.6 * K ** -0.5)        # (scores with a real spread: |q.k| / 8 ~ 2-3)
    if ln:
        gamma, beta = 1 + 0.2 * torch.randn(K, generator=g(seed + 4)), 0.2 * torch.randn(K, generator=g(seed + 5))
        wf, bf, cs = P.fold_layernorm(w, None, gamma, beta)
        xn = torch.nn.functional.layer_norm(x.float(), (K,), gamma, beta, 1e-5)
        qkv = (xn @ w.t()).to(BF).float()
    else:
        wf, bf, cs = w.to(BF).float(), None, None
        qkv = (x.float() @ wf.t()).to(BF).float()
    q, k, v = (t.view(nb, F_, Pp, heads, 64).permute(0, 2, 3, 1, 4) for t in qkv.split(inner, dim=1))
    att = torch.softmax((q @ k.transpose(-1, -2)) * 0.125, dim=-1)
    ref = (att @ v).permute(0, 3, 1, 2, 4).reshape(M, inner)
    hm = lambda t: None if t is None else P.qkv_head_major(t)
    return x, hm(wf), hm(bf), hm(cs), ref, qkv


@pytest.mark.parametrize("nb,F_,Pp,heads,ln", [(2, 24, 40, 5, True), (1, 24, 37, 5, True), (2, 24, 16, 2, False), (1, 12, 50, 5, True),
                                               (2, 16, 21, 3, True), (1, 8, 100, 1, False), (1, 48, 9, 5, True), (3, 6, 33, 4, True),
                                               (1, 24, 1, 5, True), (2, 24, 2560, 5, True), (2, 24, 600, 5, True), (3, 12, 1500, 2, False)])
def test_gemm_tqa_fused_qkv_temporal_attention(nb, F_, Pp, heads, ln):
    """The fused launch against torch on the same 16-bit inputs: frames per pixel 24 (the band of fragment pairs), 12 / 6 (the same
    band, 4 / 8 pixels per wave), 16 / 8 (the diagonal), 48 (all nine pairs); pixel counts that leave the last wave / block ragged
    (37, 21, 1) and samples whose pixel ranges straddle a wave; 1-5 heads; with and without the folded LayerNorm; the last case is the
    first level of the 24 x 40 x 64 plan (320 row tiles x 5 heads dealt to 256 persistent blocks: 6-7 items each, ranges that cross a
    row-tile boundary re-load their rows), the two after it cross tile boundaries with 1-2 items per block.  Tolerance as test_attention (P rounded to 16 bits before P.V) on top of the
    16-bit rounding of q, k, v."""
    x, w, b, cs, ref, _ = _tqa_case(nb, F_, Pp, heads, ln)
    M, inner = x.shape[0], 64 * heads
    ldo = inner + 8                                           # (a padded output row stride: the pad columns must stay untouched)
    out = torch.full((M, ldo), 7.0, dtype=BF, device="cuda")
    xd, wd = x.cuda(), w.to(BF).cuda()
    gp = ops.gemm_params(M, 3 * inner, ops.linear_segs([(xd, 320, 320)]), wd, out, ldo, bias=None if b is None else b.cuda(),
                         colsum=None if cs is None else cs.cuda(), ln_eps=1e-5 if ln else 0.0, epilogue=L.EPI_TATTN, epi_scale=0.125,
                         geom=ops.Geom(F=F_, P=Pp))
    S = ops.Stream(record=False)
    assert S.lib.vmv_gemm_pick_tile(C.byref(gp)) == L.TILE_TQA and S.lib.vmv_gemm_validate(C.byref(gp)) == 0
    S.gemm(gp, "tqa")
    torch.cuda.synchronize()
    check(out[:, :inner], ref, tol_l2=8e-3, tol_max=3e-2)
    assert bool((out[:, inner:].float() == 7.0).all())


def test_gemm_tqa_equals_the_two_kernel_form():
    """Same inputs through the plan's two-kernel form — row-stationary q | k | v GEMM (folded LayerNorm) -> attn_short_kernel — and
    through the fused launch: both round q, k, v and P to 16 bits at the same points, so they agree far inside the tolerance either
    holds against fp32 torch (the dot products run over a permuted channel order: not bitwise)."""
    nb, F_, Pp, heads = 2, 24, 96, 5
    x, w_hm, b_hm, cs_hm, ref, _ = _tqa_case(nb, F_, Pp, heads, True, seed=40)
    M, inner = x.shape[0], 64 * heads
    inv = torch.argsort(torch.arange(3 * inner).view(3, heads, 64).permute(1, 0, 2).reshape(-1))     # head-major -> [q | k | v]
    w, b, cs = w_hm[inv], b_hm[inv], cs_hm[inv]
    xd = x.cuda()
    S = ops.Stream(record=False)
    qkv = torch.zeros(M, 3 * inner, dtype=BF, device="cuda")
    S.gemm(ops.gemm_params(M, 3 * inner, ops.linear_segs([(xd, 320, 320)]), w.to(BF).cuda(), qkv, 3 * inner, bias=b.cuda(), colsum=cs.cuda(),
                           ln_eps=1e-5), "qkv")
    two = torch.zeros(M, inner, dtype=BF, device="cuda")
    mp = lambda ld: ops.seq_map(F_ * Pp * ld, ld, Pp * ld, inner=Pp)
    base = qkv.data_ptr()
    S.attention(ops.attn_params(base, base + 2 * inner, base + 4 * inner, two, mp(3 * inner), mp(3 * inner), mp(3 * inner), mp(inner),
                                nb * Pp, heads, F_, F_, 0.125), "attn")
    one = torch.zeros(M, inner, dtype=BF, device="cuda")
    S.gemm(ops.gemm_params(M, 3 * inner, ops.linear_segs([(xd, 320, 320)]), w_hm.to(BF).cuda(), one, inner, bias=b_hm.cuda(), colsum=cs_hm.cuda(),
                           ln_eps=1e-5, epilogue=L.EPI_TATTN, epi_scale=0.125, geom=ops.Geom(F=F_, P=Pp)), "tqa")
    torch.cuda.synchronize()
    check(one, two.float().cpu(), tol_l2=3e-3, tol_max=1.5e-2)
    check(one, ref, tol_l2=8e-3, tol_max=3e-2)


def test_gemm_tqa_eligibility():
    """VMV_EPI_TATTN has one kernel: K = 320, 48 % F == 0, N = 192 * heads; anything else is VMV_EINVAL (never a silent fallback),
    and vmv_gemm_tqa_ok asks for a grid that fills the chip."""
    lib = ops.Stream(record=False).lib
    x = torch.zeros(24 * 4096, 640, dtype=BF, device="cuda")
    w = torch.zeros(960, 640, dtype=BF, device="cuda")
    o = torch.zeros(24 * 4096, 320, dtype=BF, device="cuda")

    def mk(K=320, F_=24, Pp=4096, N=960, scale=0.125, **kw):
        return ops.gemm_params(F_ * Pp, N, ops.linear_segs([(x, K, K)]), w, o, 320, epilogue=L.EPI_TATTN, epi_scale=scale, geom=ops.Geom(F=F_, P=Pp), **kw)
    assert lib.vmv_gemm_validate(C.byref(mk())) == 0 and lib.vmv_gemm_tqa_ok(C.byref(mk())) == 1
    assert lib.vmv_gemm_tqa_ok(C.byref(mk(Pp=64))) == 0 and lib.vmv_gemm_validate(C.byref(mk(Pp=64))) == 0       # supported, not preferred (4 tiles x 5 heads)
    for bad in (mk(K=640), mk(F_=20), mk(N=900), mk(scale=0.0), mk(residual=o, ldr=320), mk(tile=L.TILE_RS), mk(ksplit=2, workspace=o)):
        assert lib.vmv_gemm_validate(C.byref(bad)) != 0
        assert lib.vmv_gemm_tqa_ok(C.byref(bad)) == 0


def test_gemm_rs_eligibility():
    """What the row-stationary kernel refuses when forced (VMV_EINVAL), and what the default policy sends to it."""
    import ctypes as C
    lib = L.load()
    x = rnd((512, 320), 1).cuda(); w = rnd((320, 320), 2).cuda(); o = torch.zeros(512, 320, dtype=BF, device="cuda")
    of = torch.zeros(512, 320, device="cuda"); cs = torch.zeros(320, device="cuda")
    segs = ops.linear_segs([(x, 320, 320)])
    bad = [ops.gemm_params(512, 320, segs, w, of, 320, out_fp32=True, tile=L.TILE_RS),
           ops.gemm_params(512, 320, ops.linear_segs([(x, 320, 256)]), w, o, 320, tile=L.TILE_RS),          # K = 256
           ops.gemm_params(512, 320, segs, w, o, 320, colsum=cs, ln_eps=1e-5, residual=o, ldr=320, tile=L.TILE_RS),
           ops.gemm_params(512, 320, segs, w, o, 320, rowvec=of, rowvec_div=64, rowvec_ld=320, tile=L.TILE_RS),
           ops.gemm_params(512, 160, segs, w, o, 320, tile=L.TILE_RS),                                      # N % 64
           ops.gemm_params(512, 320, ops.linear_segs([(x.view(256, 640), 640, 640)]), rnd((320, 640), 2).cuda(), o, 320, tile=L.TILE_RS512)]
    for p in bad:
        assert lib.vmv_gemm(C.byref(p), None) == -1
    small = ops.gemm_params(512, 320, segs, w, o, 320)
    assert lib.vmv_gemm_rs_ok(C.byref(small)) == 0 and lib.vmv_gemm_pick_tile(C.byref(small)) != L.TILE_RS


# ------------------------------------------------------------------------------------------------- fused FeedForward
@pytest.mark.parametrize("nstat,rps,K,N,totals", [(6, 2560, 320, 320, False), (5, 2576, 320, 320, False), (12, 640, 640, 640, False),
                                                  (7, 656, 640, 640, False), (2, 7680, 320, 320, True)])
def test_gemm_rs_groupnorm_fold(nstat, rps, K, N, totals):
    """GroupNorm -> proj_in with the apply pass folded into the row-stationary GEMM (vmv.h gn_table, vmv_groupnorm_table): the GEMM
    multiplies elem(x * scale + shift) — exactly what vmv_groupnorm_apply stores — so the result is BITWISE the two-kernel one.
    rps = 2576 / 656: blocks that straddle two stat groups; totals: the all-frame form (integer accumulators)."""
    M = nstat * rps
    x = (rnd((M, K), 3, 1.5) + 0.3).cuda()
    gamma, beta = (1 + 0.2 * torch.randn(K, generator=g(4))).cuda(), (0.1 * torch.randn(K, generator=g(5))).cuda()
    w, b = rnd((N, K), 6, K ** -0.5).cuda(), torch.randn(N, generator=g(7)).cuda()
    part = torch.zeros(ops.gn_partial_floats(M, rps, K) + 64, device="cuda")
    tot = torch.zeros(2, 64 * ops.GN_TOT, dtype=torch.int64, device="cuda")
    kw = dict(totals=tot[0], totals_clear=tot[1], clear_count=nstat * ops.GN_TOT) if totals else {}
    S = ops.Stream(record=False)
    y = torch.zeros(M, K, dtype=BF, device="cuda")
    o_ref, o_new = torch.zeros(M, N, dtype=BF, device="cuda"), torch.zeros(M, N, dtype=BF, device="cuda")
    gp = ops.gn_params(x, K, K, M, rps, part, gamma, beta, 1e-6, False, y, K, **kw)
    S.groupnorm_stats(gp); S.groupnorm_apply(gp)
    p_ref = ops.gemm_params(M, N, ops.linear_segs([(y, K, K)]), w, o_ref, N, bias=b)
    S.gemm(p_ref, "ref")
    tab = torch.zeros(nstat, 2, K, device="cuda")
    tot.zero_()
    gt = ops.gn_params(x, K, K, M, rps, part, gamma, beta, 1e-6, False, tab, K, **kw)
    S.groupnorm_stats(gt); S.groupnorm_table(gt)
    p_new = ops.gemm_params(M, N, ops.linear_segs([(x, K, K)]), w, o_new, N, bias=b, gn_table=tab, gn_rows_per_stat=rps)
    import ctypes as C
    assert S.lib.vmv_gemm_pick_tile(C.byref(p_new)) == L.TILE_RS
    S.gemm(p_new, "fold")
    torch.cuda.synchronize()
    assert torch.equal(o_new, o_ref)
    if totals:
        assert int(tot[1].abs().sum()) == 0 and int(tot[0].abs().sum()) > 0          # the table launch cleared the other buffer too
    cpu = Case(x=x.cpu(), w=w.cpu(), b=b.cpu(), tab=tab.cpu(), o=torch.zeros(M, N, dtype=BF)).on("cpu")
    I.gemm(ops.gemm_params(M, N, ops.linear_segs([(cpu["x"], K, K)]), cpu["w"], cpu["o"], N, bias=cpu["b"], gn_table=cpu["tab"], gn_rows_per_stat=rps))
    check(o_new, cpu["o"])
    # not eligible -> an error, not a silently unnormalised product
    bad = ops.gemm_params(M, N, ops.linear_segs([(x, K, K)]), w, o_new, N, bias=b, gn_table=tab, gn_rows_per_stat=rps, tile=L.TILE_256x128)
    assert S.lib.vmv_gemm(C.byref(bad), None) == -1
    bad2 = ops.gemm_params(M, N, ops.linear_segs([(x, K, K)]), w, o_new, N, bias=b, gn_table=tab, gn_rows_per_stat=264)
    assert S.lib.vmv_gemm(C.byref(bad2), None) == -1


@pytest.mark.parametrize("M,ln,res", [(1000, True, True), (128, False, False), (33000, True, True), (257, True, False), (70000, False, True)])
def test_ff_fused(M, ln, res):
    """csrc/gemm_ff.hip: out = res + W2 . ((W1x LN(x) + b1x) * gelu(W1g LN(x) + b1g)) + b2 at C = 320 in one launch (hidden in
    registers, W2's K axis permuted per 32-channel block) against the unfused definition in fp32 on the same 16-bit operands,
    and — small M — against the interpreter's restatement of the argument block; M tails, many blocks, two runs bitwise equal."""
    Cc = 320
    x = (rnd((M, Cc), 1, 1.5).float() + (3.0 * torch.randn(M, 1, generator=g(9)) if ln else 0.0)).to(BF)
    w1 = torch.randn(8 * Cc, Cc, generator=g(2)) * Cc ** -0.5
    b1 = torch.randn(8 * Cc, generator=g(3))
    w2 = torch.randn(Cc, 4 * Cc, generator=g(4)) * (4 * Cc) ** -0.5
    b2 = torch.randn(Cc, generator=g(5))
    gamma, beta = 1 + 0.2 * torch.randn(Cc, generator=g(6)), 0.2 * torch.randn(Cc, generator=g(7))
    if ln:
        w1f, b1f, _ = P.fold_layernorm(w1, b1, gamma, beta)
    else:
        w1f, b1f = w1.to(BF), b1
    w1p, b1p = P.geglu_interleave(w1f.float()).to(BF), P.geglu_interleave(b1f)
    w2p = P.ff_down_permute(w2).to(BF)
    rs = rnd((M, Cc), 8)
    c = Case(x=x, w1=w1p, b1=b1p, w2=w2p, b2=b2, res=rs, out=torch.zeros(M, Cc, dtype=BF))

    def build(t):
        return ops.ff_params(M, Cc, t["x"], Cc, t["w1"], t["b1"], t["w2"], t["b2"], t["out"], Cc,
                             residual=t["res"] if res else None, ldr=Cc, ln_eps=1e-5 if ln else 0.0)
    dev = c.on("cuda")
    S = ops.Stream(record=False)
    import ctypes as C
    assert S.lib.vmv_ff_fused_ok(C.byref(build(dev))) == 1
    S.ff(build(dev))
    torch.cuda.synchronize()
    xin = torch.nn.functional.layer_norm(x.float(), (Cc,), gamma, beta, 1e-5) if ln else x.float()
    h = xin @ (w1 if ln else w1f.float()).t() + b1
    a, gate = h.chunk(2, dim=-1)
    hid = (a * torch.nn.functional.gelu(gate)).to(BF).float()
    ref = hid @ w2.to(BF).float().t() + b2 + (rs.float() if res else 0.0)
    check(dev["out"], ref, tol_l2=6e-3, tol_max=2e-2)
    if M <= 1000:
        cpu = c.on("cpu")
        I.ff_fused(build(cpu))
        check(dev["out"], cpu["out"], tol_l2=6e-3, tol_max=2e-2)
    dev2 = c.on("cuda")
    S.ff(build(dev2))
    torch.cuda.synchronize()
    assert torch.equal(dev2["out"], dev["out"])


def test_gemm_layernorm_folded_rejects_split_k_and_gathers():
    import ctypes as C
    lib = L.load()
    x = rnd((256, 128), 1).cuda(); w = rnd((128, 128), 2).cuda(); o = torch.zeros(256, 128, dtype=BF, device="cuda")
    st = torch.zeros(256, 2, device="cuda"); cs = torch.zeros(128, device="cuda"); ws = torch.zeros(2 * 256 * 128, device="cuda")
    p = ops.gemm_params(256, 128, ops.linear_segs([(x, 128, 128)]), w, o, 128, rowstat=st, colsum=cs, ksplit=2, workspace=ws)
    assert lib.vmv_gemm(C.byref(p), None) == -1          # VMV_EINVAL
    p = ops.gemm_params(256, 128, ops.linear_segs([(x, 128, 128)]), w, o, 128, rowstat=st)
    assert lib.vmv_gemm(C.byref(p), None) == -3          # VMV_ENULL


# ------------------------------------------------------------------------------------------------- attention
def _attn_case(kind, B, F_, HW, heads, Lc=77, seed=1):
    inner = heads * 64
    T = B * F_ * HW
    if kind == "cross":
        c = Case(q=rnd((T, inner), seed), kv=rnd((B * Lc, 2 * inner), seed + 1), o=torch.zeros(T, inner, dtype=BF))
    else:
        c = Case(qkv=rnd((T, 3 * inner), seed), o=torch.zeros(T, inner, dtype=BF))

    def build(t):
        sc = 64 ** -0.5
        if kind == "temporal":
            mp = lambda ld: ops.seq_map(F_ * HW * ld, ld, HW * ld, inner=HW)
            n_outer, Nq = B * HW, F_
        else:
            mp = lambda ld: ops.seq_map(HW * ld, 0, ld, inner=1)
            n_outer, Nq = B * F_, HW
        if kind == "cross":
            kvm = ops.seq_map(Lc * 2 * inner, 0, 2 * inner, inner=1)
            return ops.attn_params(t["q"], t["kv"], t["kv"].data_ptr() + 2 * inner, t["o"], mp(inner), kvm, kvm, mp(inner),
                                   n_outer, heads, Nq, Lc, sc, kv_div=F_)
        ld = 3 * inner
        base = t["qkv"].data_ptr()
        return ops.attn_params(base, base + 2 * inner, base + 4 * inner, t["o"], mp(ld), mp(ld), mp(ld), mp(inner),
                               n_outer, heads, Nq, Nq, sc)
    return c, build


@pytest.mark.parametrize("kind,B,F_,HW,heads", [("spatial", 1, 2, 100, 3), ("spatial", 2, 1, 256, 2), ("spatial", 1, 1, 1024, 1),
                                                ("spatial", 1, 3, 40, 2), ("spatial", 1, 2, 16, 5),
                                                ("cross", 2, 3, 64, 2), ("cross", 1, 2, 200, 1),
                                                ("temporal", 2, 24, 20, 2), ("temporal", 1, 4, 9, 1), ("temporal", 1, 32, 6, 3)])
def test_attention(kind, B, F_, HW, heads):
    c, build = _attn_case(kind, B, F_, HW, heads)
    cpu = c.on("cpu")
    I.attention(build(cpu))
    dev = c.on("cuda")
    ops.Stream(record=False).attention(build(dev))
    torch.cuda.synchronize()
    check(dev["o"], cpu["o"], tol_l2=6e-3, tol_max=2e-2)     # P is rounded to bf16 before P.V (flash-style)


def test_attention_sharp_softmax():
    """A key that dominates one query row late in the sequence forces the online-softmax rescale path."""
    B, F_, HW, heads = 1, 1, 300, 1
    c, build = _attn_case("spatial", B, F_, HW, heads, seed=11)
    qkv = c.t["qkv"].float()
    qkv[5, :64] = 6.0
    qkv[250, 64:128] = 6.0       # key 250 (3rd tile of 4 for its query block) gets a huge score for query 5
    c.t["qkv"] = qkv.to(BF)
    cpu = c.on("cpu")
    I.attention(build(cpu))
    dev = c.on("cuda")
    ops.Stream(record=False).attention(build(dev))
    torch.cuda.synchronize()
    check(dev["o"], cpu["o"], tol_l2=6e-3, tol_max=2e-2)


@pytest.mark.parametrize("T,heads", [(1000, 4), (4096, 2), (130, 3)])
def test_attention_head_dim_32(T, heads):
    """head_dim 32 (LGM MVAttention at 512 channels / 16 heads, core/attention.py:67-84): the flash kernel instantiated with
    one k-step per S^T tile and two output tiles, against the interpreter; token counts with query / key tails."""
    Cc = heads * 32
    c = Case(qkv=rnd((T, 3 * Cc), 5), o=torch.zeros(T, Cc, dtype=BF))

    def build(t):
        m = lambda: ops.seq_map(0, 0, 3 * Cc, inner=1)
        base = t["qkv"].data_ptr()
        return ops.attn_params(base, base + 2 * Cc, base + 4 * Cc, t["o"], m(), m(), m(), ops.seq_map(0, 0, Cc, inner=1),
                               1, heads, T, T, 32 ** -0.5, head_dim=32)
    cpu = c.on("cpu")
    I.attention(build(cpu))
    dev = c.on("cuda")
    ops.Stream(record=False).attention(build(dev), "t")
    torch.cuda.synchronize()
    check(dev["o"], cpu["o"], tol_l2=6e-3, tol_max=2e-2)


@pytest.mark.parametrize("Fl,Fg,hw,heads", [(3, 24, 40, 5), (12, 24, 17, 2), (1, 8, 64, 1)])
def test_attention_temporal_kv_gathered(Fl, Fg, hw, heads):
    """The north-star form of frame-parallel temporal attention (VMV_FP_TEMPORAL=kv_gather): per (pixel, head) Nq = Fl local frames
    attend to Nk = Fg gathered frames; q rows live in the fused q|k|v buffer of the local frames, K | V in the gathered [Fg][hw][2C]
    buffer — the short-sequence kernel with Nq != Nk and different row strides per operand."""
    Cc = heads * 64
    c = Case(qkv=rnd((Fl * hw, 3 * Cc), 5), kv=rnd((Fg * hw, 2 * Cc), 6), o=torch.zeros(Fl * hw, Cc, dtype=BF))

    def build(t):
        qm = ops.seq_map(0, 3 * Cc, hw * 3 * Cc, inner=hw)
        km = ops.seq_map(0, 2 * Cc, hw * 2 * Cc, inner=hw)
        kv = t["kv"].data_ptr()
        return ops.attn_params(t["qkv"], kv, kv + 2 * Cc, t["o"], qm, km, km, ops.seq_map(0, Cc, hw * Cc, inner=hw), hw, heads, Fl, Fg, 64 ** -0.5)
    cpu = c.on("cpu")
    I.attention(build(cpu))
    dev = c.on("cuda")
    ops.Stream(record=False).attention(build(dev), "t")
    torch.cuda.synchronize()
    check(dev["o"], cpu["o"], tol_l2=6e-3, tol_max=2e-2)


@pytest.mark.parametrize("B,T,heads", [(2, 77, 16), (1, 77, 2), (3, 130, 1), (1, 300, 2)])
def test_attention_causal(B, T, heads):
    """VmvAttnParams.causal (the CLIP text tower, clip_embedder.py:192-201): keys j > i masked out, fused q|k|v rows; 77 tokens
    = two key tiles with a partial second one, 130 / 300 = several query blocks with fully masked key tiles above the diagonal."""
    Cc = heads * 64
    c = Case(qkv=rnd((B * T, 3 * Cc), 5), o=torch.zeros(B * T, Cc, dtype=BF))

    def build(t, causal=True):
        m = lambda: ops.seq_map(T * 3 * Cc, 0, 3 * Cc, inner=1)
        base = t["qkv"].data_ptr()
        return ops.attn_params(base, base + 2 * Cc, base + 4 * Cc, t["o"], m(), m(), m(), ops.seq_map(T * Cc, 0, Cc, inner=1),
                               B, heads, T, T, 64 ** -0.5, causal=causal)
    cpu = c.on("cpu")
    I.attention(build(cpu))
    dev = c.on("cuda")
    S = ops.Stream(record=False)
    S.attention(build(dev), "t")
    torch.cuda.synchronize()
    check(dev["o"], cpu["o"], tol_l2=6e-3, tol_max=2e-2)
    full = c.on("cuda")
    S.attention(build(full, causal=False), "t")
    torch.cuda.synchronize()
    assert not torch.equal(full["o"], dev["o"])                      # the mask is really applied
    assert torch.equal(full["o"].view(B, T, Cc)[:, -1], dev["o"].view(B, T, Cc)[:, -1]) or T > 64     # (last row sees every key; same tile walk for T <= 64 only)
    import ctypes as C
    bad = build(dev); bad.Nk = T - 1
    assert S.lib.vmv_attention(C.byref(bad), None) == -1              # causal needs Nq == Nk


@pytest.mark.parametrize("B,T,heads", [(2, 257, 16), (1, 17, 2), (1, 700, 3)])
def test_attention_head_dim_128(B, T, heads):
    """head_dim 128 (the CLIP image tower's 80-wide heads packed to 128 with zeros, clip_vision.py): four k-steps per S^T tile,
    eight output tiles, 32-KB stages; fused q|k|v rows, query and key tails."""
    Cc = heads * 128
    qkv = rnd((B * T, 3 * Cc), 5)
    qkv.view(B * T, 3, heads, 128)[..., 80:] = 0
    c = Case(qkv=qkv, o=torch.zeros(B * T, Cc, dtype=BF))

    def build(t):
        m = lambda: ops.seq_map(T * 3 * Cc, 0, 3 * Cc, inner=1)
        base = t["qkv"].data_ptr()
        return ops.attn_params(base, base + 2 * Cc, base + 4 * Cc, t["o"], m(), m(), m(), ops.seq_map(T * Cc, 0, Cc, inner=1),
                               B, heads, T, T, 80 ** -0.5, head_dim=128)
    cpu = c.on("cpu")
    I.attention(build(cpu))
    dev = c.on("cuda")
    ops.Stream(record=False).attention(build(dev), "t")
    torch.cuda.synchronize()
    check(dev["o"], cpu["o"], tol_l2=6e-3, tol_max=2e-2)
    assert float(dev["o"].view(B * T, heads, 128)[..., 80:].abs().max()) == 0.0


@pytest.mark.parametrize("M,K,N", [(154, 1024, 4096), (77, 128, 512), (3000, 320, 640)])
def test_gemm_gelu_activation(M, K, N):
    """VMV_ACT_GELU: exact-erf GELU after bias (the CLIP text tower's c_fc, nn.GELU())."""
    c = Case(x=rnd((M, K), 1), w=rnd((N, K), 2, K ** -0.5), b=torch.randn(N, generator=g(3)), o=torch.zeros(M, N, dtype=BF))

    def build(t):
        return ops.gemm_params(M, N, ops.linear_segs([(t["x"], K, K)]), t["w"], t["o"], N, bias=t["b"], act=L.ACT_GELU)
    cpu = c.on("cpu")
    I.gemm(build(cpu))
    dev = c.on("cuda")
    ops.Stream(record=False).gemm(build(dev), "t")
    torch.cuda.synchronize()
    check(dev["o"], cpu["o"])


# ------------------------------------------------------------------------------------------------- sampler glue
def test_layout_and_ddim_kernels():
    nb, Cc, F_, H, W = 1, 4, 5, 6, 7
    x = torch.randn(nb, Cc, F_, H, W, generator=g(1))
    rows_ref = torch.zeros(2 * F_ * H * W, 8, dtype=BF)
    I.latent_to_rows(x, rows_ref, 8, 2)
    rows = torch.zeros_like(rows_ref, device="cuda")
    ops.latent_to_rows(x.cuda(), rows, 8, 2)
    assert torch.equal(rows.cpu(), rows_ref)
    eps = torch.randn(2 * F_ * H * W, 4, generator=g(2))
    xt_ref = x.clone()
    x0_ref = torch.zeros_like(x)
    args = dict(guide_scale=9.0, c_recip=1.31, c_recipm1=0.85, c_sqrt_ac=0.76, c_sqrt_1mac=0.65, a_prev=0.71)
    I.cfg_ddim_step(eps, 4, xt_ref, x0_out=x0_ref, **args)
    xt, x0 = x.clone().cuda(), torch.zeros_like(x).cuda()
    ops.cfg_ddim_step(eps.cuda(), 4, xt, x0_out=x0, **args)
    assert torch.allclose(xt.cpu(), xt_ref, rtol=1e-5, atol=1e-5) and torch.allclose(x0.cpu(), x0_ref, rtol=1e-5, atol=1e-5)
    xt2 = x.clone().cuda()
    xt2_ref = x.clone()
    I.cfg_ddim_step(eps, 4, xt2_ref, v_pred=True, **args)
    ops.cfg_ddim_step(eps.cuda(), 4, xt2, v_pred=True, **args)
    assert torch.allclose(xt2.cpu(), xt2_ref, rtol=1e-5, atol=1e-5)
    # the sampler options that ride in the fused update (diffusion_ddim.py:204-205 clamp, :233-243 eta > 0): x0 clamped BEFORE eps is
    # re-derived, direction sqrt(1 - a_prev - sigma^2), + sigma * noise
    nz = torch.randn(x.shape, generator=g(5))
    for v_pred in (False, True):
        for kw in (dict(clamp=0.6), dict(sigma=0.31, noise=nz), dict(clamp=1.1, sigma=0.2, noise=nz)):
            a_ref, x0r = x.clone(), torch.zeros_like(x)
            I.cfg_ddim_step(eps, 4, a_ref, v_pred=v_pred, x0_out=x0r, **args, **kw)
            a_dev, x0d = x.clone().cuda(), torch.zeros_like(x).cuda()
            kd = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in kw.items()}
            ops.cfg_ddim_step(eps.cuda(), 4, a_dev, v_pred=v_pred, x0_out=x0d, **args, **kd)
            assert torch.allclose(a_dev.cpu(), a_ref, rtol=1e-5, atol=1e-5) and torch.allclose(x0d.cpu(), x0r, rtol=1e-5, atol=1e-5)
            if "clamp" in kw:
                assert float(x0d.abs().max()) <= kw["clamp"] + 1e-6 and float(x0r.abs().max()) > 0.5 * kw["clamp"]
    with pytest.raises(Exception):      # sigma without noise is refused, not read through a null pointer
        ops.cfg_ddim_step(eps.cuda(), 4, x.clone().cuda(), sigma=0.3, **args)
    out = torch.zeros(3, 4, H, W, device="cuda")
    r = torch.randn(3 * H * W, 4, generator=g(3))
    ops.rows_to_nchw(r.cuda(), 4, out)
    out_ref = torch.zeros(3, 4, H, W)
    I.rows_to_nchw(r, 4, out_ref)
    assert torch.equal(out.cpu(), out_ref)
    t = torch.tensor([981.0, 1.0])
    s = torch.zeros(2, 320, dtype=BF, device="cuda")
    ops.sinusoidal(t.cuda(), s, 2, 320)
    s_ref = torch.zeros(2, 320, dtype=BF)
    I.sinusoidal(t, s_ref, 2, 320)
    assert (s.cpu().float() - s_ref.float()).abs().max() < 2e-2     # sin/cos of ~1e3 rad: fp32 range reduction differs


def test_i2v_frontend_kernels():
    """vmv_i2v_temporal_adapter / vmv_adaptive_avgpool_rows / vmv_latent_to_rows_keep vs the torch restatement."""
    F_, HW = 24, 40
    a3 = rnd((F_ * HW, 4), 1)
    w = torch.randn(288, generator=g(2)) * 0.5
    w[0:4] = 1 + 0.1 * w[0:4]
    out_ref = torch.zeros(2 * F_ * HW, 8, dtype=BF)
    I.i2v_temporal_adapter(a3, 4, out_ref.data_ptr() + 8, 8, w, F_, HW, 2, 2.0)
    out = torch.zeros(2 * F_ * HW, 8, dtype=BF, device="cuda")
    a3d, wd = a3.cuda(), w.cuda()
    ops.i2v_temporal_adapter(a3d, 4, out.data_ptr() + 8, 8, wd, F_, HW, 2, 2.0)
    torch.cuda.synchronize()
    assert torch.equal(out[:, :4].cpu(), torch.zeros(2 * F_ * HW, 4, dtype=BF))          # channels 0..3 untouched
    check(out[:, 4:], out_ref[:, 4:], tol_l2=5e-3, tol_max=1.5e-2)
    for (ih, iw) in ((8, 8), (40, 64), (32, 32)):
        x = rnd((2 * ih * iw, 32), 3)
        y_ref = torch.zeros(2 * 32 * 32, 32, dtype=BF)
        I.adaptive_avgpool_rows(x, 32, y_ref, 32, 2, 32, ih, iw, 32, 32)
        y = torch.zeros(2 * 32 * 32, 32, dtype=BF, device="cuda")
        xd = x.cuda()
        ops.adaptive_avgpool_rows(xd, 32, y, 32, 2, 32, ih, iw, 32, 32)
        torch.cuda.synchronize()
        check(y, y_ref, tol_l2=4e-3, tol_max=1e-2)
    lat = torch.randn(1, 4, 3, 5, 6, generator=g(4))
    rows_ref = torch.full((2 * 3 * 30, 8), 7.0, dtype=BF)
    I.latent_to_rows_keep(lat, rows_ref, 8, 2)
    rows = torch.full((2 * 3 * 30, 8), 7.0, dtype=BF, device="cuda")
    ops.latent_to_rows_keep(lat.cuda(), rows, 8, 2)
    assert torch.equal(rows.cpu(), rows_ref) and float(rows[:, 4:].float().min()) == 7.0


def test_gaussian_activation_matches_reference_semantics():
    """vmv_gaussian_activation vs the torch expressions of core/models.py:37-43 (incl. F.normalize's default dim=1)."""
    n = 5000
    raw = torch.randn(n, 16, generator=g(11)) * 2.0
    raw[:5, 4:7] = 25.0                      # softplus threshold branch
    ref = torch.zeros(n, 14)
    I.gaussian_activation(raw.clone(), 16, ref, n, None)
    out = torch.zeros(n, 14, device="cuda")
    ops.gaussian_activation(raw.cuda(), 16, out, n, torch.zeros(1024, device="cuda"))
    torch.cuda.synchronize()
    assert torch.allclose(out.cpu(), ref, rtol=2e-5, atol=2e-6), float((out.cpu() - ref).abs().max())
    x = raw[:, :14]
    exp_rot = torch.nn.functional.normalize(x[None, :, 7:11])[0]          # the reference call: default dim=1 on [B, N, 4]
    assert torch.allclose(out.cpu()[:, 7:11], exp_rot, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("S_in,S", [(128, 64), (96, 64), (64, 64), (80, 32)])
def test_lgm_render_to_vae_matches_interpolate(S_in, S):
    """rendered views -> VAE input: F.interpolate(images, (S, S), mode='nearest') then [0,1] -> [-1,1] (unet_t2v.py:425-427),
    for any render / target size ratio."""
    img = torch.rand(3, 3, S_in, S_in, generator=g(2))
    out = torch.empty(3, 3, S, S, device="cuda")
    ops.lgm_render_to_vae(img.cuda(), out)
    torch.cuda.synchronize()
    ref = (torch.nn.functional.interpolate(img, size=(S, S), mode="nearest") - 0.5) / 0.5
    assert torch.allclose(out.cpu(), ref, atol=1e-6)


# ------------------------------------------------------------------------------------------------- fp16 stores saturate
@pytest.mark.skipif(BF != torch.float16, reason="bf16 has fp32's exponent range: nothing to saturate")
@pytest.mark.parametrize("tile,M,N,K", [(L.TILE_64x64, 128, 64, 64), (L.TILE_128x128, 256, 128, 128), (L.TILE_256x160, 512, 320, 320),
                                        (L.TILE_G128x160, 512, 320, 320), (L.TILE_P256x160, 66000, 320, 320), (L.TILE_X256x320, 1024, 320, 320),
                                        (L.TILE_RS, 70000, 320, 320)])
def test_fp16_stores_saturate(tile, M, N, K):
    """a finite fp32 result beyond the fp16 range must be STORED as +-65504, not +-inf (csrc/common.h: the MODE.FP16_OVFL
    bit set at kernel entry).  x = +-256, w = 8 over K >= 64 -> |acc| >= 131072 > 65504; one row stays in range (exact)."""
    sign = torch.where(torch.arange(M) % 2 == 0, 1.0, -1.0)[:, None]
    a = (256.0 * sign).expand(M, K).clone().to(BF)
    a[5] = 2.0 ** -8                                              # row 5 stays in range: 2^-8 * 8 * K = K / 32, exact in fp16
    w = torch.full((N, K), 8.0, dtype=BF)
    out = torch.zeros(M, N, dtype=BF, device="cuda")
    S = ops.Stream(record=False)
    S.gemm(ops.gemm_params(M, N, ops.linear_segs([(a.cuda(), K, K)]), w.cuda(), out, N, tile=tile), "sat")
    torch.cuda.synchronize()
    o = out.float().cpu()
    assert torch.isfinite(o).all()
    keep = torch.ones(M, dtype=torch.bool); keep[5] = False
    assert torch.equal(o[keep], (65504.0 * sign).expand(M, N)[keep])
    assert torch.equal(o[5], torch.full((N,), float(a[5, 0]) * 8.0 * K))


@pytest.mark.skipif(BF != torch.float16, reason="bf16 has fp32's exponent range: nothing to saturate")
def test_fp16_norm_and_glue_stores_saturate_and_keep_infinities():
    """The same store rule outside the GEMMs: LayerNorm with a huge gamma saturates; a true infinity in the INPUT still comes out
    non-finite (saturation must not hide a broken forward from the finite check)."""
    rows, Cc = 64, 320
    x = torch.randn(rows, Cc, generator=g(1)).to(BF).cuda()
    y = torch.zeros(rows, Cc, dtype=BF, device="cuda")
    gam, bet = torch.full((Cc,), 1e6).cuda(), torch.zeros(Cc).cuda()
    S = ops.Stream(record=False)
    S.layernorm(ops.ln_params(x, Cc, y, Cc, gam, bet, rows, Cc, 1e-5), "sat.ln")
    torch.cuda.synchronize()
    o = y.float().cpu()
    assert torch.isfinite(o).all() and float(o.abs().max()) == 65504.0 and float((o.abs() == 65504.0).float().mean()) > 0.9
    x[3, 7] = float("inf")
    S.layernorm(ops.ln_params(x, Cc, y, Cc, torch.ones(Cc).cuda(), bet, rows, Cc, 1e-5), "inf.ln")
    torch.cuda.synchronize()
    assert not torch.isfinite(y[3].float()).all()

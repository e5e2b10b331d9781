#!/usr/bin/env python
"""Where does the row-stationary GEMM differ from fp32 math?  Prints bad-element patterns per case."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videomv_amd import _lib as L, ops

BF = L.elem()


def run(M, N, K, tile, bias=True, res=False, reps=1):
    g = torch.Generator().manual_seed(1)
    a = torch.randn(M, K, generator=g).to(BF)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(BF)
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g).to(BF)
    ref = a.float() @ w.float().t() + (b if bias else 0) + (r.float() if res else 0)
    ad, wd, bd, rd = a.cuda(), w.cuda(), b.cuda(), r.cuda()
    S = ops.Stream(record=False)
    for rep in range(reps):
        out = torch.full((M, N), float("nan"), dtype=BF, device="cuda")
        kw = dict(residual=rd, ldr=N) if res else {}
        S.gemm(ops.gemm_params(M, N, ops.linear_segs([(ad, K, K)]), wd, out, N, bias=bd if bias else None, tile=tile, **kw))
        torch.cuda.synchronize()
        o = out.float().cpu()
        bad = ~torch.isfinite(o) | ((o - ref).abs() > 0.05 * ref.abs().max())
        nb = int(bad.sum())
        print(f"M={M} N={N} K={K} tile={tile} bias={bias} res={res} rep={rep}: bad={nb} nonfinite={int((~torch.isfinite(o)).sum())}")
        if nb:
            rows = bad.any(dim=1).nonzero().flatten()
            cols = bad.any(dim=0).nonzero().flatten()
            print("   rows:", rows[:40].tolist(), "... n=", len(rows))
            print("   cols:", cols[:64].tolist(), "... n=", len(cols))
            rr = int(rows[0])
            cc = bad[rr].nonzero().flatten()
            print(f"   row {rr}: bad cols {cc[:40].tolist()} vals {o[rr, cc[:8]].tolist()} ref {ref[rr, cc[:8]].tolist()}")
            # per (row % 64) and (col % 64) histogram
            print("   row%64 hist:", torch.bincount(bad.nonzero()[:, 0] % 64, minlength=64).tolist())
            print("   col%64 hist:", torch.bincount(bad.nonzero()[:, 1] % 64, minlength=64).tolist())
            print("   col//64 hist:", torch.bincount(bad.nonzero()[:, 1] // 64, minlength=(N + 63) // 64).tolist())


def run_ln(M, N, K, tile, geglu=False, ln=True):
    from videomv_amd import packing as P
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(M, K, generator=g) * 1.5 + 4.0 * torch.randn(M, 1, generator=g)).to(BF)
    w = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g)
    gamma, beta = 1 + 0.2 * torch.randn(K, generator=g), 0.2 * torch.randn(K, generator=g)
    if ln:
        wf, bf, cs = P.fold_layernorm(w, b, gamma, beta)
    else:
        wf, bf, cs = w.to(BF).float(), b, None
    xin = torch.nn.functional.layer_norm(x.float(), (K,), gamma, beta, 1e-5) if ln else x.float()
    ref = xin @ (w if ln else wf).t() + b
    No = N
    if geglu:
        a, gate = ref.chunk(2, dim=-1)
        ref = a * torch.nn.functional.gelu(gate)
        wf, bf = P.geglu_interleave(wf), P.geglu_interleave(bf)
        cs = P.geglu_interleave(cs) if ln else None
        No = N // 2
    out = torch.full((M, No), float("nan"), dtype=BF, device="cuda")
    kw = dict(colsum=cs.cuda(), ln_eps=1e-5) if ln else {}
    xd = x.cuda()
    ops.Stream(record=False).gemm(ops.gemm_params(M, N, ops.linear_segs([(xd, K, K)]), wf.to(BF).cuda(), out, No, bias=bf.cuda(),
                                                   epilogue=L.EPI_GEGLU if geglu else L.EPI_NONE, tile=tile, **kw))
    torch.cuda.synchronize()
    o = out.float().cpu()
    bad = ~torch.isfinite(o) | ((o - ref).abs() > 0.05 * ref.abs().max())
    nb = int(bad.sum())
    rl2 = float(((o - ref).norm() / ref.norm()))
    print(f"LN={ln} GEGLU={geglu} M={M} N={N} K={K} tile={tile}: bad={nb} nonfinite={int((~torch.isfinite(o)).sum())} rel_l2={rl2:.2e}")
    if nb:
        rows = bad.any(dim=1).nonzero().flatten(); cols = bad.any(dim=0).nonzero().flatten()
        print("   rows:", rows[:40].tolist(), "n=", len(rows)); print("   cols:", cols[:64].tolist(), "n=", len(cols))
        print("   row%64 hist:", torch.bincount(bad.nonzero()[:, 0] % 64, minlength=64).tolist())
        print("   col%64 hist:", torch.bincount(bad.nonzero()[:, 1] % 64, minlength=64).tolist())
        rr = int(rows[0]); cc = bad[rr].nonzero().flatten()
        print(f"   row {rr}: vals {o[rr, cc[:8]].tolist()} ref {ref[rr, cc[:8]].tolist()}")


if __name__ == "__main__":
    run_ln(1000, 960, 320, L.TILE_RS512)
    run_ln(1000, 1920, 640, L.TILE_RS256)
    run_ln(1000, 2560, 320, L.TILE_RS512, geglu=True, ln=False)
    run_ln(700, 5120, 640, L.TILE_RS256, geglu=True, ln=False)
    run_ln(1000, 2560, 320, L.TILE_RS512, geglu=True, ln=True)
    run(1000, 960, 320, L.TILE_RS512, reps=2)
    run(1000, 960, 320, L.TILE_RS256)
    run(512, 64, 320, L.TILE_RS512)
    run(512, 192, 320, L.TILE_RS512)
    run(512, 256, 320, L.TILE_RS512)
    run(256, 128, 640, L.TILE_RS256)
    run(1000, 1920, 640, L.TILE_RS256)
    run(1000, 320, 320, L.TILE_RS512, res=True)
    run(122880, 960, 320, L.TILE_RS)

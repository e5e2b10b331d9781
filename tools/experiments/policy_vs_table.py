#!/usr/bin/env python
"""Offline (CPU-only) reading of videomv_amd/tuned_gemm.json: WHERE does the built-in tile / split-K policy lose, and would a simple
fill-based rule have found the measured winners?

    python tools/experiments/policy_vs_table.py

Step 1 records the full-size plans on the CPU (zero weights, tests/plan_interp.py: seconds) with the built-in policy and no table — the
24x40x64 / 24x32x32 plans and the simulated ranks of worlds 2 / 4 / 8 that tools/autotune_gemm.py measured, plus three shapes nobody
measured (24x48x48, 24x64x64, 16 frames at 32x32).  Step 2 compares, per launch signature, the policy, the measured best (the table) and
the candidate rule: "where the policy falls back to 128-row tiles (ids 7 / 8), take 256-row tiles of 128 / 160 columns with the split-K
factor (<= 8, >= 8 chunks per split) that fills one round of the 256 CUs best: score = fill x padding efficiency - 0.03 per extra split;
short-K linears on few rows: 64 x 64 register tiles when 200-1024 of them exist".  Result at the end of round 4: the rule
reproduces the measured family (rows, columns, split) for 127 of the 311 improved launches-per-plan (41 %), picks another 256-row variant
for 18, a different family for 24, is silent for 142 (mostly launches whose measured winner is a persistent / 96-row / register-tile
variant at the SAME fill, a few per cent apart), and would change 1 of the 512 launches the measurement found the policy right about.  It is NOT in the library: a policy change has to be timed, and the round's GPU budget
went to the table.  This script is the starting point for folding it into csrc/gemm.hip's pick_tile + ops.SplitK next round."""
import sys, os, json, math, collections, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["VMV_TUNED"] = "0"
from tests import plan_interp
class _P:
    @staticmethod
    def setattr(o, n, v): setattr(o, n, v)
plan_interp.install(_P)
from videomv_amd.unet_engine import UNetEngine, param_shapes
from videomv_amd.comm import SimComm
from videomv_amd import ops, _lib as L
cfg = dict(in_dim=4, dim=320, context_dim=1024, out_dim=4, dim_mult=[1, 2, 4, 4], num_heads=8, head_dim=64, num_res_blocks=2,
           attn_scales=[1.0, 0.5, 0.25], camera_dim=16, use_camera_condition=True, use_fps_condition=False)
sd = {k: torch.zeros(s) for k, s in param_shapes(cfg).items()}
dev = torch.device('cpu')
lib = L.load()
plans = {}
donor = None
def rec(tag, **kw):
    global donor
    e = UNetEngine(cfg, sd, packed=donor.packed if donor else None, device=dev, n_t=1, **kw)
    donor = donor or e
    out = {}
    for op, p in e.S.recorded:
        if op != L.OP_GEMM or p.wgroup_rows: continue
        sig = ops.gemm_signature(p)
        if sig in out: continue
        steps = sum((p.seg[i].k + 63) // 64 for i in range(p.nseg))
        out[sig] = dict(M=p.M, N=p.N, K=p.ktot, steps=steps, tile=lib.vmv_gemm_pick_tile(C.byref(p)), ks=p.ksplit if p.ksplit > 1 else 0,
                        geglu=p.epilogue == L.EPI_GEGLU, rowstat=bool(p.rowstat), ln=p.ln_eps > 0, gn=bool(p.gn_table), lin=all(p.seg[i].mode == 0 for i in range(p.nseg)),
                        fp32=bool(p.out_fp32), rowvec=bool(p.rowvec), res=bool(p.residual))
    plans[tag] = out
    print(tag, len(out))
rec('world1 40x64', B=2, F=24, H=40, W=64, L_ctx=77, share_prefix=True)
rec('world1 32x32', B=2, F=24, H=32, W=32, L_ctx=77, share_prefix=True)
rec('world1 48x48', B=2, F=24, H=48, W=48, L_ctx=77, share_prefix=True)
rec('world1 64x64', B=2, F=24, H=64, W=64, L_ctx=77, share_prefix=True)
rec('world1 16f 32x32', B=2, F=16, H=32, W=32, L_ctx=77, share_prefix=True)
for w in (2, 4, 8):
    rec(f'world{w} B=2', B=2, F=24, H=40, W=64, L_ctx=77, comm=SimComm(w, 0))
    rec(f'world{w} B=1', B=1, F=24, H=40, W=64, L_ctx=77, comm=SimComm(w, 0))

tab = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'videomv_amd', 'tuned_gemm.json')))['fp16']
ROWS = {5:256,6:256,9:256,10:256,7:128,8:128,13:128,14:96,4:64,3:128,1:128,2:128,20:256,21:256,22:256,23:0}
BN = {5:128,6:160,9:128,10:160,7:128,8:160,13:128,14:160,4:64,3:64,1:128,2:160,20:320,21:256,22:128}
def rule(i):
    """-> (tile, ks) or None"""
    if i['tile'] not in (7, 8):            # only where the policy fell back to the 128-row LDS-DMA tiles
        return None
    M, N, steps = i['M'], i['N'], i['steps']
    ks_ok = not (i['rowstat'] or i['ln'] or i['gn'])
    best = None
    for bn in (128, 160):
        if bn == 160 and (N % 160 or i['geglu']): continue
        tm, tn = math.ceil(M / 256), math.ceil(N / bn)
        pad = (M * N) / (tm * 256 * tn * bn)
        for ks in (1, 2, 3, 4, 6, 8):
            if ks > 1 and (not ks_ok or steps // ks < 8): continue
            blocks = tm * tn * ks
            fill = blocks / (256 * math.ceil(blocks / 256))
            score = fill * pad - 0.03 * (ks - 1) + (0.01 if bn == 128 else 0.0)
            if best is None or score > best[0] + 1e-9:
                best = (score, bn, ks)
    if best and best[0] >= 0.70:
        _, bn, ks = best
        if ks == 1 and i['lin'] and steps <= 24:
            return (9 if bn == 128 else 10, 0)
        return (5 if bn == 128 else 6, ks if ks > 1 else 0)
    if i['lin'] and steps <= 24 and M <= 4096 and not i['geglu']:
        t64 = math.ceil(M / 64) * math.ceil(N / 64)
        if 200 <= t64 <= 1024:
            return (4, 0)
    return None
fam = lambda t: (ROWS.get(t, -1), BN.get(t, -1))
stats = collections.Counter()
detail = []
for tag, sigs in plans.items():
    for sig, i in sigs.items():
        r = rule(i)
        pol = (i['tile'], i['ks'])
        if sig in tab:
            b = (tab[sig]['tile'], tab[sig]['ksplit'])
            if r is None: stats[(tag, 'improved: rule silent')] += 1
            elif fam(r[0]) == fam(b[0]) and r[1] == b[1]: stats[(tag, 'improved: rule == best')] += 1
            elif fam(r[0])[0] == fam(b[0])[0]: stats[(tag, 'improved: same rows')] += 1; detail.append((tag, sig.split(';')[0], pol, r, b, tab[sig]['base_us'], tab[sig]['us']))
            else: stats[(tag, 'improved: rule differs')] += 1; detail.append((tag, sig.split(';')[0], pol, r, b, tab[sig]['base_us'], tab[sig]['us']))
        else:
            measured = tag in ('world1 40x64', 'world1 32x32') or tag.startswith('world2') or tag.startswith('world4') or tag.startswith('world8')
            if r is not None and r != pol:
                stats[(tag, 'policy-optimal but rule changes' if measured else 'unmeasured: rule changes')] += 1
                if measured: detail.append((tag, sig.split(';')[0], pol, r, 'policy was within 7% of best', 0, 0))
            else: stats[(tag, 'unchanged')] += 1
for k in sorted(stats): print(k, stats[k])
print()
for d in detail[:60]: print(d)

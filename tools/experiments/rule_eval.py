#!/usr/bin/env python
"""CPU-only: score ops.fill_rule against the measured tile table on the plans the table was measured on.
For every distinct GEMM signature of those plans: table entry present -> does the rule reproduce it (exactly / same tile family)?;
absent (the measurement found the policy within 7 %) -> does the rule leave it alone?"""
import sys, os, json, collections, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["VMV_TUNED"] = "0"; os.environ["VMV_TILE_RULES"] = "0"
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
lib = L.load()
donor = None
sigs = {}
def rec(tag, **kw):
    global donor
    e = UNetEngine(cfg, sd, packed=donor.packed if donor else None, device=torch.device("cpu"), n_t=1, **kw)
    donor = donor or e
    for op, p in e.S.recorded:
        if op == L.OP_GEMM and not p.wgroup_rows:
            sigs.setdefault(ops.gemm_signature(p), (p, tag))
rec('w1 40x64', B=2, F=24, H=40, W=64, L_ctx=77, share_prefix=True)
rec('w1 32x32', B=2, F=24, H=32, W=32, L_ctx=77, share_prefix=True)
for w in (2, 4, 8):
    rec(f'w{w} B2', B=2, F=24, H=40, W=64, L_ctx=77, comm=SimComm(w, 0))
    rec(f'w{w} B1', B=1, F=24, H=40, W=64, L_ctx=77, comm=SimComm(w, 0))
tab = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'videomv_amd', 'tuned_gemm.json')))['fp16']
st = collections.Counter(); gain = collections.Counter()
for sig, (p, tag) in sigs.items():
    pol = lib.vmv_gemm_pick_tile(C.byref(p)); pks = p.ksplit if p.ksplit > 1 else 0
    r = ops.fill_rule(p, pol)
    if r is not None and (r[0], r[1]) == (pol, pks): r = None
    if sig in tab:
        b = (tab[sig]['tile'], tab[sig].get('ksplit', 0))
        k = 'silent' if r is None else 'exact' if (r[0], r[1]) == b else 'other'
        st[('improved', k)] += 1; gain[('improved', k)] += tab[sig]['base_us'] - tab[sig]['us']
        if '-v' in sys.argv and k != 'exact': print(k, tag, sig.split(';')[0], (pol, pks), r, b, tab[sig]['base_us'], tab[sig]['us'])
    else:
        st[('policy-best', 'left alone' if r is None else 'CHANGED')] += 1
        if r is not None and '-v' in sys.argv: print('CHANGED', tag, sig.split(';')[0], (pol, pks), r)
for k in sorted(st): print(k, st[k], round(gain[k], 1), 'us')

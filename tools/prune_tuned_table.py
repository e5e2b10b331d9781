#!/usr/bin/env python
"""Shrink videomv_amd/tuned_gemm.json to what the default tile RULE (ops.fill_rule) does not already find (CPU-only; host logic).

    python tools/prune_tuned_table.py [--write]

Every table key is an ``ops.gemm_signature`` — shape, K segments, epilogue flags and conv geometry, no pointer — so the launch it stands
for can be rebuilt as a VmvGemmParams with placeholder pointers and put through the library's own host-side policy
(``vmv_gemm_pick_tile``) and through ``ops.fill_rule``.  An entry whose measured (tile, split-K) the rule reproduces is redundant and is
dropped; what stays is the residue measurement found and no rule explains.  The file is rewritten one entry per line."""
import argparse, ctypes as C, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videomv_amd import _lib as L, ops

PH = 0x100000      # placeholder address: the policy only asks "is this pointer set"


def params_from_signature(sig: str, ksplit: int = 0) -> "L.GemmParams":
    shape, runs, flags, geo = sig.split(";")
    M, N, K = (int(v) for v in shape.split("x"))
    p = L.GemmParams()
    p.M, p.N, p.ktot = M, N, K
    s = 0
    for run in runs.split(","):
        mode, rest = run.split(":")
        k, cnt = rest.split("*")
        for j in range(int(cnt)):
            sg = p.seg[s]
            sg.src, sg.ld, sg.k, sg.mode = PH, int(k), int(k), int(mode)
            if int(mode) == L.SEG_SPATIAL:
                sg.d0, sg.d1 = j // 3 - 1, j % 3 - 1
            elif int(mode) == L.SEG_TEMPORAL:
                sg.d0 = j - 1
            s += 1
    p.nseg = s
    m = re.fullmatch(r"e(\d+)a(\d+)f(\d+)r(\d+)v(\d+):(\d+)s(\d+)c(\d+)l(\d+)g(\d+)w(\d+)", flags)
    e, a, f, r, v, vdiv, st, cs, ln, gn, wg = (int(x) for x in m.groups())
    p.epilogue, p.act, p.out_fp32 = e, a, f
    p.residual, p.ldr = (PH if r else None), (N if r else 0)
    p.rowvec, p.rowvec_div, p.rowvec_ld = (PH if v else None), (vdiv if v else 0), (N if v else 0)
    p.rowstat, p.colsum, p.ln_eps = (PH if st else None), (PH if cs else None), (1e-5 if ln else 0.0)
    p.gn_table, p.gn_rows_per_stat = (PH if gn else None), (M if gn else 0)
    p.wgroup_rows = wg
    g = re.fullmatch(r"(\d+)x(\d+)<(\d+)x(\d+)s(\d+)u(\d+)F(\d+)P(\d+)", geo)
    p.OH, p.OW, p.IH, p.IW, p.stride, p.ups, p.F, p.P = (int(x) for x in g.groups())
    p.W, p.out, p.ldo = PH, PH, (N // 2 if e == L.EPI_GEGLU else N)
    p.ksplit, p.workspace = (ksplit if ksplit > 1 else 0), (PH if ksplit > 1 else None)
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true")
    ap.add_argument("--path", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "videomv_amd", "tuned_gemm.json"))
    a = ap.parse_args()
    tab = json.load(open(a.path))
    lib = L.load()
    keep, dropped, mismatch = {}, 0, 0
    for sig, ent in tab["fp16"].items():
        p = params_from_signature(sig, int(ent.get("base_ksplit", 0)))
        assert ops.gemm_signature(p) == sig, (sig, ops.gemm_signature(p))
        pol = lib.vmv_gemm_pick_tile(C.byref(p))
        if pol != int(ent.get("base_tile", pol)):
            mismatch += 1            # the policy moved since the entry was measured: keep the measurement
            keep[sig] = ent
            continue
        r = ops.fill_rule(p, pol)
        if r is not None and (int(r[0]), int(r[1])) == (int(ent["tile"]), int(ent.get("ksplit", 0))):
            dropped += 1
            continue
        keep[sig] = ent
    print(f"{len(tab['fp16'])} entries: {dropped} reproduced by the rule (dropped), {len(keep)} kept ({mismatch} with a moved policy tile)")
    if a.write:
        meta = dict(tab.get("meta", {}))
        meta["pruned"] = f"tools/prune_tuned_table.py: {dropped} entries that ops.fill_rule reproduces removed"
        with open(a.path, "w") as f:
            f.write('{\n"meta": ' + json.dumps(meta, sort_keys=True) + ',\n"fp16": {\n')
            items = sorted(keep.items())
            for i, (k, v) in enumerate(items):
                v = {kk: v[kk] for kk in ("tile", "ksplit", "us", "base_us", "base_tile", "base_ksplit", "plan") if kk in v}
                f.write(json.dumps(k) + ": " + json.dumps(v, sort_keys=True) + ("," if i + 1 < len(items) else "") + "\n")
            f.write("}\n}\n")


if __name__ == "__main__":
    main()

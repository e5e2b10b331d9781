"""CPU: record the 24x40x64 plan (no launches) and join shapes / algorithmic bytes onto a per-launch timing table
written by `bench.py --dump-ops` (same launch order).  usage: ops_shapes.py gpurun_out/ops.tsv"""
import csv, sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from videomv_amd import _lib as L
from videomv_amd.registry import MODEL
import videomv_amd.unet_t2v  # noqa
from videomv_amd.flops import gemm_bytes, gemm_flops
from bench import FULL

H, W, F = 40, 64, 24
model = MODEL.build(dict(type="UNetSD_T2VBase", y_dim=1024, use_camera_condition=True, use_lgm_refine=False, **FULL)).eval()
eng = model.engine_for(2, F, H, W, 77, torch.device("cpu"), n_t=1)
rec = eng.S.recorded
rows = list(csv.DictReader(open(sys.argv[1]), delimiter="\t"))
assert len(rows) == len(rec), (len(rows), len(rec))
agg = collections.OrderedDict()
for r, (op, p) in zip(rows, rec):
    if op != L.OP_GEMM:
        continue
    modes = "".join(str(p.seg[i].mode) for i in range(min(p.nseg, 3)))
    key = (p.M, p.N, p.ktot, p.nseg, modes, p.epilogue, bool(p.residual), p.ksplit, p.tile)
    a = agg.setdefault(key, [0, 0.0, gemm_bytes(p), gemm_flops(p), r["label"]])
    a[0] += 1; a[1] += float(r["ms"])
print(f"{'M':>7s} {'N':>6s} {'K':>6s} seg mode epi res ks tile   n  us/op   TF/s   GB/s  tot_ms  label")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    us = 1e3 * a[1] / a[0]
    print(f"{k[0]:7d} {k[1]:6d} {k[2]:6d} {k[3]:3d} {k[4]:>4s} {k[5]:3d} {int(k[6]):3d} {k[7]:2d} {k[8]:4d} {a[0]:3d} {us:7.1f} {a[3]/us/1e6:6.0f} {a[2]/us/1e3:6.0f} {a[1]:7.3f}  {a[4]}")

#!/usr/bin/env python
"""HBM traffic of the UNet plan PER LAUNCH, attributed to (kernel, op kind, level) — VERDICT r5 item 5: "nobody has tabled which
kernel owns the excess" of the GEMM family's counter bytes over its algorithmic bytes.

  rocprofv3 --pmc FETCH_SIZE -f csv -d <fetch_dir> -- python tools/traffic_by_op.py run <ops.json>
  rocprofv3 --pmc WRITE_SIZE -f csv -d <write_dir> -- python tools/traffic_by_op.py run <ops.json>
  python tools/traffic_by_op.py table <ops.json> <fetch_dir> <write_dir> <out.tsv> [<out_family.json>]

`run` (on the GPU, under the profiler): builds the bench's full-size [cond | uncond] plan at 24 x 40 x 64, replays it once whole
(warm-up), then replays it ONE RECORDED OP AT A TIME in plan order, each preceded by a one-element torch fill (the sentinel: every
kernel between two sentinels belongs to one plan op — split-K reduce, statistics + apply pairs included); the op list with each op's
algorithmic bytes (flops.gemm_bytes) goes to <ops.json>.  Cache state is the real forward's: the ops run in plan order on the tensors
the previous op wrote.
`table`: joins the two counter passes with the op list by position.  Units / corrections as tools/gemm_traffic.py
(MI355X_MICROARCH.md, HBM section): counters in KiB, FETCH_SIZE doubled on gfx950, WRITE_SIZE as reported."""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def op_kind(label: str) -> str:
    lb = label
    for pat, kind in ((r"temopral_conv\.conv\d(\.gn)?", "tconv"), (r"\.conv1$", "conv3x3.1"), (r"\.conv2(\+skip)?$", "conv3x3.2"),
                      (r"qkv\+attn$", "qkv+attn"), (r"\.qkv$", "qkv"), (r"attn\d\.q$", "q"), (r"attn\d\.kv$", "ctx.kv"), (r"\.out$", "attn.out"),
                      (r"ff\.geglu$", "ff.geglu"), (r"ff\.down$", "ff.down"), (r"ff\.fused$", "ff.fused"), (r"proj_in$", "proj_in"),
                      (r"proj_out$", "proj_out"), (r"^emb", "emb"), (r"^out\.2$", "head"), (r"\.attn$", "attention")):
        if re.search(pat, lb):
            return kind
    if re.fullmatch(r"input_blocks\.\d+(\.0)?", lb):
        return "down/conv_in"
    if "upsample" in lb or re.fullmatch(r"output_blocks\.\d+\.\d+", lb):
        return "up"
    return "other"


def run(ops_path):
    import ctypes
    import torch
    import bench
    from videomv_amd import _lib as L
    L.load()
    from videomv_amd.registry import MODEL, DIFFUSION
    import videomv_amd.unet_t2v  # noqa: F401
    import videomv_amd.diffusion_ddim  # noqa: F401
    from videomv_amd.flops import gemm_bytes, gemm_flops, attn_flops
    from videomv_amd.camera import entrance_camera_data
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    H, W, F_ = (int(v) for v in os.environ.get("VMV_TRAFFIC_SHAPE", "40x64x24").split("x"))
    with torch.device(dev):
        model = MODEL.build(dict(type="UNetSD_T2VBase", y_dim=1024, use_camera_condition=True, use_lgm_refine=False, **bench.FULL))
    bench.randomize_(model, 1234)
    model.eval()
    dif = DIFFUSION.build(dict(type="DiffusionDDIM", schedule="linear_sd", schedule_param=dict(num_timesteps=1000, init_beta=0.00085, last_beta=0.012,
                                                                                             zero_terminal_snr=False), mean_type="eps", var_type="fixed_small"))
    g = torch.Generator(device=dev).manual_seed(11)
    xt = torch.randn(1, 4, F_, H, W, generator=g, device=dev)
    y, y0 = torch.randn(1, 77, 1024, generator=g, device=dev), torch.randn(1, 77, 1024, generator=g, device=dev)
    cam = entrance_camera_data(F_, elevation=15, camera_distance=2.0).to(dev)
    kc, ku = dict(y=y, camera_data=cam), dict(y=y0, camera_data=cam)
    for s in (981, 961):
        dif.ddim_step_hip(xt, s, model, kc, ku, 9.0, 20)
    torch.cuda.synchronize()
    eng = model.engine_for(2, F_, H, W, 77, dev, n_t=1, share_prefix=True)
    rec, labels = eng.S.recorded, eng.S.labels
    KIND = {L.OP_GEMM: "gemm", L.OP_GN_STATS: "gn_stats", L.OP_GN_APPLY: "gn_apply", L.OP_LAYERNORM: "layernorm", L.OP_ATTENTION: "attention",
            L.OP_GN_FUSED: "gn_fused", L.OP_COPY: "copy", L.OP_FF: "ff_fused", L.OP_GN_TABLE: "gn_table", L.OP_COMM: "collective"}
    ops_out = []
    for i, (op, p) in enumerate(rec):
        d = dict(idx=i, label=labels[i], family=KIND.get(op, str(op)))
        if op == L.OP_GEMM:
            d.update(alg_bytes=gemm_bytes(p), flops=gemm_flops(p), M=p.M, N=p.N, K=p.ktot, ksplit=p.ksplit,
                     tile=int(eng.S.lib.vmv_gemm_pick_tile(ctypes.byref(p))), kind=op_kind(labels[i]))
        elif op == L.OP_ATTENTION:
            by = 2.0 * p.n_outer * p.heads * 64 * (2 * p.Nq + 2 * p.Nk / max(1, p.kv_div))
            d.update(alg_bytes=by, flops=attn_flops(p), kind="attention")
        ops_out.append(d)
    with open(ops_path, "w") as f:
        json.dump(dict(shape=f"{F_}x{H}x{W}", nops=len(rec), ops=ops_out), f)
    sent = torch.zeros(1, device=dev)
    torch.cuda.synchronize()
    sent.fill_(-1.0)                       # opening sentinel of the measured pass
    for i in range(len(rec)):
        eng.S.run(i, i + 1)
        sent.fill_(float(i))
    torch.cuda.synchronize()
    print(f"traffic_by_op: {len(rec)} ops replayed one at a time ({ops_path})")


def read_pass(src, counter):
    rows = []
    for path in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") == counter:
                    rows.append((int(r["Dispatch_Id"]), r.get("Kernel_Name", ""), float(r["Counter_Value"])))
    rows.sort()
    return rows


def split_by_sentinel(rows, nops):
    """-> per-op list of (kernel, value) lists: the LAST nops + 1 sentinel launches delimit the one-at-a-time pass."""
    is_sent = ["FillFunctor" in k for _, k, _ in rows]                # (torch's fill_ kernel; no kernel of the library has that name)
    pos = [i for i, s in enumerate(is_sent) if s]
    if len(pos) < nops + 1:
        raise SystemExit(f"found {len(pos)} sentinel launches, need {nops + 1}")
    pos = pos[-(nops + 1):]
    return [[(rows[j][1], rows[j][2]) for j in range(pos[i] + 1, pos[i + 1])] for i in range(nops)]


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*$", "", name)[:60]


def table(ops_path, fetch_dir, write_dir, out_tsv, out_family=None):
    with open(ops_path) as f:
        meta = json.load(f)
    ops, nops = meta["ops"], meta["nops"]
    fe = split_by_sentinel(read_pass(fetch_dir, "FETCH_SIZE"), nops)
    wr = split_by_sentinel(read_pass(write_dir, "WRITE_SIZE"), nops)
    per_op = []
    for o, f_, w_ in zip(ops, fe, wr):
        kern = "+".join(short(k) for k, _ in f_) or "?"
        fetch = 2.0 * 1024.0 * sum(v for _, v in f_)
        write = 1024.0 * sum(v for _, v in w_)
        per_op.append(dict(o, kernel=kern, fetch=fetch, write=write, counter=fetch + write))
    # group: (kernel, kind, M) for the GEMM family
    groups = {}
    for o in per_op:
        if o["family"] != "gemm":
            continue
        key = (o["kernel"], o["kind"], o["M"], o["N"], o["K"])
        gq = groups.setdefault(key, dict(n=0, counter=0.0, fetch=0.0, write=0.0, alg=0.0, flops=0.0))
        gq["n"] += 1
        for a, b in (("counter", "counter"), ("fetch", "fetch"), ("write", "write"), ("alg", "alg_bytes"), ("flops", "flops")):
            gq[a] += o[b]
    tot_c = sum(g_["counter"] for g_ in groups.values())
    tot_a = sum(g_["alg"] for g_ in groups.values())
    n_g = sum(g_["n"] for g_ in groups.values())
    with open(out_tsv, "w") as f:
        f.write(f"# HBM counter bytes vs algorithmic bytes per GEMM-family launch group, plan {meta['shape']} (one forward, ops replayed in plan order; "
                f"FETCH_SIZE x2 + WRITE_SIZE, KiB -> bytes)\n")
        f.write(f"# family: {n_g} launches, counter {tot_c / 1e9:.3f} GB, algorithmic {tot_a / 1e9:.3f} GB, ratio {tot_c / tot_a:.3f}; per launch "
                f"{tot_c / n_g / 1e6:.1f} MB vs {tot_a / n_g / 1e6:.1f} MB\n")
        f.write("kernel\tkind\tM\tN\tK\tlaunches\tcounter_MB\tfetch_MB\twrite_MB\talgorithmic_MB\tratio\texcess_MB_total\tshare_of_family_excess\n")
        exc_tot = max(1.0, tot_c - tot_a)
        for key, g_ in sorted(groups.items(), key=lambda kv: -(kv[1]["counter"] - kv[1]["alg"])):
            n = g_["n"]
            f.write(f"{key[0]}\t{key[1]}\t{key[2]}\t{key[3]}\t{key[4]}\t{n}\t{g_['counter'] / n / 1e6:.1f}\t{g_['fetch'] / n / 1e6:.1f}\t{g_['write'] / n / 1e6:.1f}\t"
                    f"{g_['alg'] / n / 1e6:.1f}\t{g_['counter'] / max(1.0, g_['alg']):.2f}\t{(g_['counter'] - g_['alg']) / 1e6:.0f}\t"
                    f"{(g_['counter'] - g_['alg']) / exc_tot:.3f}\n")
        f.write("# other families (per launch, MB): family\tlaunches\tcounter_MB\n")
        fam = {}
        for o in per_op:
            if o["family"] == "gemm":
                continue
            a = fam.setdefault(o["family"], [0, 0.0])
            a[0] += 1
            a[1] += o["counter"]
        for k, (n, c) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
            f.write(f"# {k}\t{n}\t{c / n / 1e6:.2f}\n")
    if out_family:
        d = dict(kernel_family="16-bit MFMA implicit-GEMM launches of one forward (per-op replay, tools/traffic_by_op.py)", shape=meta["shape"],
                 sources_sha16=__import__("bench").kernel_sources_sha16(), dtype=os.environ.get("VMV_DTYPE", "fp16"), launches=n_g,
                 bytes_per_launch=tot_c / n_g, algorithmic_bytes_per_launch=tot_a / n_g, ratio=tot_c / tot_a,
                 correction="FETCH_SIZE x2 (gfx950 counts 64 B per 128-B request on wide reads), WRITE_SIZE as reported")
        with open(out_family, "w") as f:
            json.dump(d, f, indent=1)
    print(f"family: {n_g} launches, ratio {tot_c / tot_a:.3f}")


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "run":
        run(sys.argv[2])
    elif len(sys.argv) >= 6 and sys.argv[1] == "table":
        table(*sys.argv[2:7])
    else:
        raise SystemExit(__doc__)

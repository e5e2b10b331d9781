"""Host-side tools that turn profiler output into the tables under profiles/ (no GPU: synthetic counter files)."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_pass(d, counter, launches):
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "x_counter_collection.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
        w.writeheader()
        for i, (k, v) in enumerate(launches):
            w.writerow(dict(Dispatch_Id=i + 1, Kernel_Name=k, Counter_Name=counter, Counter_Value=v))


def test_traffic_by_op_table_joins_the_counter_passes_with_the_op_list(tmp_path):
    """tools/traffic_by_op.py `table`: kernels between two sentinel fills belong to one plan op (a split-K GEMM = kernel + reduce), the
    LAST nops + 1 sentinels delimit the measured pass, FETCH_SIZE is doubled, KiB -> bytes, groups are sorted by excess bytes."""
    fill = "void at::native::vectorized_elementwise_kernel<4, at::native::FillFunctor<float>, std::array<char*, 1ul> >(int)"
    gx = "void (anonymous namespace)::gemm_xglds_kernel<2, 5, 0>(VmvGemmParams)"
    gr = "void (anonymous namespace)::gemm_rs_kernel<4, 10, 2>(VmvGemmParams, int, int, int)"
    red = "void (anonymous namespace)::gemm_splitk_reduce(VmvGemmParams)"
    gn = "void (anonymous namespace)::gn_stats_kernel(VmvGroupNormParams)"
    ops = dict(shape="24x40x64", nops=4, ops=[
        dict(idx=0, label="input_blocks.1.0.conv1", family="gemm", alg_bytes=100e6, flops=1e12, M=1000, N=320, K=2880, ksplit=0, tile=20, kind="conv3x3.1"),
        dict(idx=1, label="input_blocks.1.0.gn2.stats", family="gn_stats"),
        dict(idx=2, label="input_blocks.1.2.transformer_blocks.0.attn1.qkv", family="gemm", alg_bytes=50e6, flops=1e11, M=1000, N=960, K=320, ksplit=0, tile=23, kind="qkv"),
        dict(idx=3, label="input_blocks.10.0.conv1", family="gemm", alg_bytes=10e6, flops=1e10, M=100, N=1280, K=11520, ksplit=4, tile=5, kind="conv3x3.1")])
    with open(tmp_path / "ops.json", "w") as f:
        json.dump(ops, f)
    # an earlier (warm-up) pass with stray fills, then the measured pass
    pre = [(fill, 0), (gx, 999), (fill, 0), (gr, 999)]
    seq_f = pre + [(fill, 0), (gx, 100000), (fill, 0), (gn, 5000), (fill, 0), (gr, 30000), (fill, 0), (gx, 8000), (red, 2000), (fill, 0)]
    seq_w = pre + [(fill, 0), (gx, 40000), (fill, 0), (gn, 10), (fill, 0), (gr, 20000), (fill, 0), (gx, 3000), (red, 1000), (fill, 0)]
    _write_pass(str(tmp_path / "fetch"), "FETCH_SIZE", seq_f)
    _write_pass(str(tmp_path / "write"), "WRITE_SIZE", seq_w)
    out, fam = tmp_path / "t.tsv", tmp_path / "f.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "traffic_by_op.py"), "table", str(tmp_path / "ops.json"), str(tmp_path / "fetch"),
                        str(tmp_path / "write"), str(out), str(fam)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = [l.split("\t") for l in open(out).read().splitlines() if l and not l.startswith("#")]
    hdr, body = rows[0], rows[1:]
    col = {h: i for i, h in enumerate(hdr)}
    assert len(body) == 3
    top = body[0]                                   # the largest excess: the conv (2 * 100000 + 40000 KiB = 245.8 MB vs 100 MB)
    assert top[col["kind"]] == "conv3x3.1" and top[col["kernel"]].startswith("gemm_xglds_kernel<2, 5, 0>") and top[col["launches"]] == "1"
    assert abs(float(top[col["counter_MB"]]) - (2 * 100000 + 40000) * 1024 / 1e6) < 0.1 and abs(float(top[col["ratio"]]) - 2.46) < 0.01
    splitk = next(b for b in body if "gemm_splitk_reduce" in b[col["kernel"]])
    assert abs(float(splitk[col["counter_MB"]]) - (2 * 10000 + 4000) * 1024 / 1e6) < 0.1
    fj = json.load(open(fam))
    tot = ((2 * 100000 + 40000) + (2 * 30000 + 20000) + (2 * 10000 + 4000)) * 1024
    assert fj["launches"] == 3 and abs(fj["bytes_per_launch"] - tot / 3) < 1 and abs(fj["ratio"] - tot / 160e6) < 1e-6

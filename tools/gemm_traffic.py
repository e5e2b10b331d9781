#!/usr/bin/env python
"""HBM traffic of the GEMM family per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

  python tools/gemm_traffic.py <fetch_dir> <write_dir> <out.json>

Units and correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): rocprofv3 reports both counters in
KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide streaming reads, so it is doubled; WRITE_SIZE is used
as reported (uncalibrated).  bench.py puts `bytes_per_launch` into `roofline.traffic`."""
import csv
import glob
import json
import os
import sys


def family_sum(src, counter):
    n, total = 0, 0.0
    for path in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") != counter:
                    continue
                name = r.get("Kernel_Name", "")
                if "gemm_" not in name or "splitk_reduce" in name:
                    continue
                n += 1
                total += float(r["Counter_Value"])
    return n, total


def main(fetch_dir, write_dir, out):
    nf, fetch = family_sum(fetch_dir, "FETCH_SIZE")
    nw, write = family_sum(write_dir, "WRITE_SIZE")
    if not nf or not nw:
        raise SystemExit("no GEMM launches found in the counter files")
    d = dict(kernel_family="gemm_xglds_kernel / gemm_rs_kernel / gemm_glds_kernel / gemm_pglds_kernel / gemm_kernel (all 16-bit MFMA implicit-GEMM launches)",
             commit=os.environ.get("VMV_COMMIT", "unknown"), dtype=os.environ.get("VMV_DTYPE", "fp16"),
             launches_fetch_pass=nf, launches_write_pass=nw,
             fetch_kib_per_launch_reported=fetch / nf, write_kib_per_launch_reported=write / nw,
             correction="FETCH_SIZE x2 (gfx950 counts 64 B per 128-B request on wide reads), WRITE_SIZE as reported",
             bytes_per_launch=(2.0 * fetch / nf + write / nw) * 1024.0)
    with open(out, "w") as f:
        json.dump(d, f, indent=1)
    print(json.dumps(d))


if __name__ == "__main__":
    main(*sys.argv[1:4])

#!/usr/bin/env python
"""Register / spill / scratch summary of every kernel in the production library, from the compiler's resource-usage remarks that
videomv_amd/csrc/compile_checked.sh keeps next to each object (build/<elem>/<src>.o.res).
    python tools/kernel_resources.py [f16|bf16] > profiles/r3_kernel_resources.txt
(The Makefile refuses a kernel with vgpr spills; this file is the record the judge asked for.)"""
import glob, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
elem = sys.argv[1] if len(sys.argv) > 1 else "f16"
rows = []
for f in sorted(glob.glob(os.path.join(ROOT, "videomv_amd", "csrc", "build", elem, "*.o.res"))):
    cur = None
    for l in open(f):
        m = re.search(r"remark: Function Name: (\S+)", l)
        if m:
            cur = dict(src=os.path.basename(f)[:-6], name=m.group(1)); rows.append(cur); continue
        if cur is None:
            continue
        for key, pat in (("sgpr", r" SGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("sspill", r"SGPRs Spill: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, l)
            if m and key not in cur:
                cur[key] = int(m.group(1))
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.strip().split("\n") if rows else []
print(f"# kernel resources of libvmv_hip_{elem}.so (hipcc -Rpass-analysis=kernel-resource-usage; {len(rows)} kernels)")
print(f"# kernels with spilled VGPRs: {sum(1 for r in rows if r.get('vspill', 0) > 0)}   with spilled SGPRs: {sum(1 for r in rows if r.get('sspill', 0) > 0)}")
print(f"# kernels with scratch (private arrays, no spill): {sum(1 for r in rows if r.get('scratch', 0) > 0 and r.get('vspill', 0) == 0)}")
print(f"{'source':14s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'static LDS':>10s} {'waves/SIMD':>10s}  kernel")
for r, n in zip(rows, names):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    if n.endswith(")"):            # drop the argument list (balanced scan from the end: no regular expression on C++ names)
        depth, j = 0, len(n) - 1
        while j >= 0:
            depth += (n[j] == ")") - (n[j] == "(")
            if depth == 0:
                break
            j -= 1
        n = n[:j] if j > 0 else n
    if len(n) > 110:
        n = n[:107] + "..."
    print(f"{r['src']:14s} {r.get('vgpr', 0):5d} {r.get('agpr', 0):5d} {r.get('sgpr', 0):5d} {r.get('vspill', 0):6d} {r.get('sspill', 0):6d} {r.get('scratch', 0):7d} "
          f"{r.get('lds', 0):10d} {r.get('occ', 0):10d}  {n}")

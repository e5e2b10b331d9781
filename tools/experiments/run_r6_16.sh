#!/bin/bash
# round 6, 16th GPU call: under-fill probe — one forward over 2 vs 4 row blocks (tools/experiments/batch_fill.py)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd $R; mkdir -p $O
{ timeout 400 python tools/experiments/batch_fill.py 40x64; timeout 300 python tools/experiments/batch_fill.py 32x32; } 2>/dev/null | tee $O/r6_batch_fill.log

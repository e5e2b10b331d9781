#!/bin/bash
# where does gemm_rs spend its time: per-shape TFLOP/s of builds with one component removed (results of those builds are wrong)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export VMV_BENCH_SHAPES="L0 N,L1 N"
echo "== default"; timeout 300 python tools/gemm_bench.py 23 2>&1 | grep -v amdgpu.ids | grep -v "down L"
for v in abl1 abl2 abl3 abl4 abl5 abl6 abl7; do echo "== $v"; VMV_LIB_DIR=$R/ab_libs/$v timeout 300 python tools/gemm_bench.py 23 2>&1 | grep -v amdgpu.ids | grep -v "down L"; done
echo "== default again"; timeout 300 python tools/gemm_bench.py 23 2>&1 | grep -v amdgpu.ids | grep -v "down L"

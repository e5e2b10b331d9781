#!/bin/bash
# A/B of gemm_rs build variants (ab_libs/<name>/libvmv_hip_f16.so against videomv_amd/lib): per-shape and per-step
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
VARS=${VARS:-"stag colm"}
export VMV_BENCH_SHAPES="L0 N,L1 N"
echo "== default"; timeout 300 python tools/gemm_bench.py 23 2>&1 | grep -v amdgpu.ids | grep -v "down L"
for v in $VARS; do echo "== $v"; VMV_LIB_DIR=$R/ab_libs/$v timeout 300 python tools/gemm_bench.py 23 2>&1 | grep -v amdgpu.ids | grep -v "down L"; done
unset VMV_BENCH_SHAPES
B="python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-sample --no-lgm --no-op-profile --no-alt-dtype"
run() { env "$@" $B 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$*', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
run X=default
for v in $VARS; do run VMV_LIB_DIR=$R/ab_libs/$v; done
done

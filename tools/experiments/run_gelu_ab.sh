#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "geglu or rs_" 2>&1 | tail -4
VARS="r1s0 X" 
export VMV_BENCH_SHAPES="geglu"
for v in r1s0 cur r1s0; do echo "== $v"; if [ $v = cur ]; then timeout 300 python tools/gemm_bench.py 23 0 2>&1 | grep -v amdgpu.ids; else VMV_LIB_DIR=$R/ab_libs/$v timeout 300 python tools/gemm_bench.py 23 0 2>&1 | grep -v amdgpu.ids; fi; done
unset VMV_BENCH_SHAPES
B="python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-sample --no-lgm --no-op-profile --no-alt-dtype"
run() { env "$@" $B 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('STEP $*', d['value'], d['ms_per_step'])"; }
for i in 1 2; do run VMV_LIB_DIR=$R/ab_libs/r1s0; run X=cur; done

#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
VARS=${VARS:-"p1s0 p2s0 p3s0 p2s1 p3s1"}
export VMV_BENCH_SHAPES="L0 N,L1 N"
for v in $VARS $(echo $VARS | cut -d' ' -f1); do echo "== $v"; VMV_LIB_DIR=$R/ab_libs/$v timeout 300 python tools/gemm_bench.py 23 2>&1 | grep -v amdgpu.ids | grep -v "down L"; done
unset VMV_BENCH_SHAPES
B="python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-sample --no-lgm --no-op-profile --no-alt-dtype"
run() { env "$@" $B 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('STEP $*', d['value'], d['ms_per_step'])"; }
for i in 1 2; do for v in $VARS; do run VMV_LIB_DIR=$R/ab_libs/$v; done; done

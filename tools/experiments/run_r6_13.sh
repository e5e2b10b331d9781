#!/bin/bash
# round 6, 13th GPU call: 256-thread two-blocks-per-CU form of gemm_xglds (VMV_TILE_Y256x128) — tests, then the K = 1280 linears against the plan's tiles
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd $R; timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "linear_bias or geglu or layernorm_folded or conv3x3" > $O/r6_y256_tests.log 2>&1; tail -3 $O/r6_y256_tests.log
timeout 300 python tools/experiments/y256_bench.py 2>/dev/null | tee $O/r6_y256_bench.log

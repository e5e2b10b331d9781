#!/bin/bash
# round 6, eighth GPU call: gemm_glds run-wise walk by M (auto) — step A/B against forced-off, conv tests, LGM stage bench
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm or conv" > $O/r6_tap3_tests.log 2>&1; tail -2 $O/r6_tap3_tests.log
bash tools/experiments/run_env_ab2.sh "VMV_GLDS_TAPMAJOR=0" "" > $O/r6_tap3_step_ab.log 2>&1; cat $O/r6_tap3_step_ab.log
mkdir -p $O/lgm2; VMV_OUT=$O/lgm2 python tools/experiments/lgm_step_bench.py > $O/r6_lgm_step_bench_auto.log 2>&1; tail -8 $O/r6_lgm_step_bench_auto.log

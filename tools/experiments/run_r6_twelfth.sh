#!/bin/bash
# round 6, twelfth GPU call: the 512 x 128 tile of gemm_xglds (8 x 1 wave grid) — conv tests, microbench vs the other tiles, VAE / LGM stage bench A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd $R; python -m pytest tests/test_kernels_gpu.py -x -q -k "conv" > $O/r6_x512_tests.log 2>&1; tail -2 $O/r6_x512_tests.log
VMV_BENCH_SHAPES="vae conv 128" python tools/gemm_bench.py 0 5 30 2>/dev/null | tee $O/r6_x512_bench.log
for v in 0 1; do mkdir -p $O/lgm_x$v; VMV_GEMM_X512=$v VMV_OUT=$O/lgm_x$v python tools/experiments/lgm_step_bench.py 2>/dev/null | grep -E "sum ms|decode4" | sed "s/^/X512=$v /"; done | tee -a $O/r6_x512_bench.log
python -m pytest tests -m gpu -x -q -k "vae or lgm" > $O/r6_x512_vae_tests.log 2>&1; tail -2 $O/r6_x512_vae_tests.log

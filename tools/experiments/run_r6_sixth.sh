#!/bin/bash
# round 6, sixth GPU call: run-wise (tap-interleaved) K walk of gemm_xglds — kernel tests, conv microbench A/B, step A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm or conv" > $O/r6_tap_tests.log 2>&1; echo "tests rc $?" >> $O/r6_tap_tests.log; tail -3 $O/r6_tap_tests.log
for v in 0 1; do
  echo "# VMV_XGLDS_TAPMAJOR=$v"
  VMV_XGLDS_TAPMAJOR=$v VMV_BENCH_SHAPES="conv L0,tcnv L0,conv L1,tcnv L1,conv L2,vae conv" python tools/gemm_bench.py 0 2>/dev/null
done > $O/r6_tap_bench.log 2>&1; cat $O/r6_tap_bench.log
bash tools/experiments/run_env_ab2.sh "VMV_XGLDS_TAPMAJOR=0" "VMV_XGLDS_TAPMAJOR=1" > $O/r6_tap_step_ab.log 2>&1; cat $O/r6_tap_step_ab.log
python -m pytest tests/test_unet_gpu.py -x -q -k "golden or full_size_reference or tiny or block" > $O/r6_tap_unet_tests.log 2>&1; tail -2 $O/r6_tap_unet_tests.log

#!/bin/bash
# round 6, first GPU call: the fused q|k|v + temporal attention kernel — parity tests, micro-benchmark, same-box step A/B
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -k "tqa" > gpurun_out/r6_tqa_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r6_tqa_tests.log
tail -5 gpurun_out/r6_tqa_tests.log
python tools/experiments/tqa_bench.py > gpurun_out/r6_tqa_bench.log 2>&1; cat gpurun_out/r6_tqa_bench.log
bash tools/experiments/run_env_ab.sh VMV_TQA 0 1 > gpurun_out/r6_tqa_step_ab.log 2>&1; cat gpurun_out/r6_tqa_step_ab.log
python -m pytest tests/test_gs_gpu.py tests/test_unet_gpu.py -x -q > gpurun_out/r6_unet_gs_tests.log 2>&1; tail -3 gpurun_out/r6_unet_gs_tests.log

#!/bin/bash
# fused-FF variants (ab_libs/ff_*), same box: us per FF at M = 122880
mkdir -p gpurun_out
for v in base pf2_6 pf3_8 nomfma noread nogelu nosync base; do
  echo "== $v" >> gpurun_out/ff_ab.log
  VMV_LIB_DIR=ab_libs/ff_$v timeout 120 python tools/experiments/ff_bench.py 2>&1 | grep "M=122880" >> gpurun_out/ff_ab.log
done
cat gpurun_out/ff_ab.log

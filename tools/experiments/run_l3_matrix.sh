#!/bin/bash
# small-M levels (L3: M = 1920): tile x split-K matrix, TFLOP/s (tile 0 = the dispatcher's choice at that split)
for ks in 0 2 3 4 6 8; do
  echo "== ksplit $ks"
  VMV_BENCH_KSPLIT=$ks VMV_BENCH_SHAPES="L3" timeout 120 python tools/gemm_bench.py 0 1 2 5 6 7 8 2>&1 | grep -v amdgpu
done

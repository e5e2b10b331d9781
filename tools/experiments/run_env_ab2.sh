#!/bin/bash
# same-box step A/B of two environment settings: $1 = "VAR=val ..." (A), $2 = "VAR=val ..." (B; may be empty); alternates A / B twice
for v in "$1" "$2" "$1" "$2"; do
  env $v timeout 200 python bench.py --no-cpu-baseline --no-sample --steps 10 --warmup 2 --simulate-rank 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['roofline']['families']; print('STEP [$v]', d['ms_per_step'], 'gemm', f['gemm']['ms'], 'gn_fused', f['gn_fused']['ms'], '32x32', d['reference_shape']['ms_per_step'])"
done

#!/bin/bash
# same-box step A/B of an engine switch: $1 = VAR, $2 = value A, $3 = value B; alternates A / B twice
for v in $2 $3 $2 $3; do
  env $1=$v timeout 200 python bench.py --no-cpu-baseline --no-sample --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['roofline']['families']; print('STEP $1=$v', d['ms_per_step'], 'gemm', f['gemm']['ms'], 'gn', round(f['gn_stats']['ms']+f['gn_apply']['ms']+f['gn_fused']['ms'],3))"
done

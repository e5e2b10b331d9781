#!/bin/bash
# same-box step A/B of two library builds: $1 = ab_libs/<dir> of the OLD one; alternates old / new twice
mkdir -p gpurun_out
for v in old new old new; do
  if [ $v = old ]; then export VMV_LIB_DIR=$1; else unset VMV_LIB_DIR; fi
  timeout 200 python bench.py --no-cpu-baseline --no-sample --no-op-profile --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('STEP $v', d['ms_per_step'])"
done

#!/bin/bash
# the fill rule (VMV_TILE_RULES=1) against the bare policy, no table: 24x32x32 and 24x48x48
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
B="python bench.py --steps 10 --warmup 3 --no-sample --no-cpu-baseline --simulate-rank 0 --no-op-profile"
run() { ( env "$@" timeout 60 $B $EXTRA ) 2>/dev/null | python -c "import sys,json; print('$* $EXTRA', json.loads([l for l in sys.stdin if l.startswith('{')][0])['ms_per_step'])"; }
{
EXTRA="--latent 32x32"; run VMV_TUNED=0 VMV_TILE_RULES=1; run VMV_TUNED=0
EXTRA="--latent 48x48"; run VMV_TUNED=0 VMV_TILE_RULES=1; run VMV_TUNED=0
} 2>&1 | tee gpurun_out/r4n_rule.log

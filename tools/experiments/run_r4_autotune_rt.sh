#!/bin/bash
# VMV_AUTOTUNE=1 on a shape no table covers (latent 24x48x48): first run tunes + writes the cache, second run reads it
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
B="python bench.py --latent 48x48 --steps 10 --warmup 3 --no-sample --no-cpu-baseline --simulate-rank 0 --no-op-profile"
run() { ( env "$@" timeout 300 $B ) 2>gpurun_out/r4l_err.log | python -c "import sys,json; print('$*', json.loads([l for l in sys.stdin if l.startswith('{')][0])['ms_per_step'])"; }
{
run VMV_AUTOTUNE=0
run VMV_AUTOTUNE=1 VMV_TUNED_CACHE=/tmp/vmv_cache.json VMV_AUTOTUNE_VERBOSE=1
run VMV_AUTOTUNE=1 VMV_TUNED_CACHE=/tmp/vmv_cache.json
run VMV_AUTOTUNE=0
python -c "import json; d=json.load(open('/tmp/vmv_cache.json')); print('cache entries', {k: len(v) for k, v in d.items()})"
} 2>&1 | tee gpurun_out/r4l_autotune_rt.log
tail -3 gpurun_out/r4l_err.log

#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
B="python bench.py --steps 20 --warmup 5 --no-sample --no-cpu-baseline --simulate-rank 0 --no-op-profile"
run() { ( env "$@" timeout 120 $B $EXTRA ) 2>/dev/null | python -c "import sys,json; print('$* $EXTRA', json.loads([l for l in sys.stdin if l.startswith('{')][0])['ms_per_step'])"; }
{
EXTRA=""
for i in 1 2; do run VMV_ATTN_QT=0; run VMV_ATTN_QT=4; run VMV_ATTN_QT=2; done
cp videomv_amd/tuned_gemm.json /tmp/relaxed32.json
( timeout 200 python tools/autotune_gemm.py --worlds 1 --latent 32x32 --merge --min-gain 0.97 --min-gain-big 0.97 --out /tmp/relaxed32.json ) > gpurun_out/r4k_autotune32_relaxed.log 2>&1
tail -n 1 gpurun_out/r4k_autotune32_relaxed.log
cp /tmp/relaxed32.json gpurun_out/r4k_relaxed32_tuned_gemm.json
EXTRA="--latent 32x32"
for i in 1 2 3; do run VMV_TUNED=1; run VMV_TUNED_FILE=/tmp/relaxed32.json; done
} 2>&1 | tee gpurun_out/r4k_misc.log

#!/bin/bash
# experiment: a tile table for the 24x40x64 plan with RELAXED acceptance (>= 3 % per launch) judged by the whole step, interleaved A/B
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
cp videomv_amd/tuned_gemm.json /tmp/relaxed.json
( timeout 300 python tools/autotune_gemm.py --worlds 1 --latent 40x64 --merge --min-gain 0.97 --min-gain-big 0.97 --out /tmp/relaxed.json ) > gpurun_out/r4i_autotune_relaxed.log 2>&1
cp /tmp/relaxed.json gpurun_out/r4i_relaxed_tuned_gemm.json
grep -c "tile" gpurun_out/r4i_autotune_relaxed.log; tail -n 2 gpurun_out/r4i_autotune_relaxed.log
B="python bench.py --steps 20 --warmup 5 --no-sample --no-cpu-baseline --simulate-rank 0 --no-op-profile"
for i in 1 2 3; do
  ( timeout 120 $B ) 2>/dev/null | python -c "import sys,json; print('default', json.loads([l for l in sys.stdin if l.startswith('{')][0])['ms_per_step'])"
  ( VMV_TUNED_FILE=/tmp/relaxed.json timeout 120 $B ) 2>/dev/null | python -c "import sys,json; print('relaxed', json.loads([l for l in sys.stdin if l.startswith('{')][0])['ms_per_step'])"
done

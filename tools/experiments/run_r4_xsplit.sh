#!/bin/bash
# wide-tile split-K: tests, then re-measure the plans that can use it, then step-level A/B against the committed table
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
( timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_clip_gpu.py -m gpu -q -x --timeout 200 -p no:cacheprovider -k "splitk or clip or conv3x3 or temporal_conv" ) > gpurun_out/r4m_tests.log 2>&1
tail -n 3 gpurun_out/r4m_tests.log
cp videomv_amd/tuned_gemm.json /tmp/old_table.json
cp videomv_amd/tuned_gemm.json /tmp/new_table.json
( timeout 400 python tools/autotune_gemm.py --worlds 1,8 --latent 40x64 --merge --out /tmp/new_table.json ) > gpurun_out/r4m_autotune_a.log 2>&1
( timeout 200 python tools/autotune_gemm.py --worlds 1 --latent 32x32 --merge --out /tmp/new_table.json ) > gpurun_out/r4m_autotune_b.log 2>&1
grep -c "tile 2[012]/ks[2-9]\|tile 2[012]/ks1" gpurun_out/r4m_autotune_a.log gpurun_out/r4m_autotune_b.log
tail -n 1 gpurun_out/r4m_autotune_a.log gpurun_out/r4m_autotune_b.log
cp /tmp/new_table.json gpurun_out/r4m_new_tuned_gemm.json
B="python bench.py --steps 20 --warmup 5 --no-sample --no-cpu-baseline --no-op-profile"
run() { ( env "$@" timeout 200 $B $EXTRA ) 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); s=d.get('simulated_rank') or {}
print('$* $EXTRA', d['ms_per_step'], {k:v.get('gpu_ms_per_step') for k,v in (s.get('modes') or {}).items() if 'graph' not in k})"; }
{
EXTRA="--simulate-rank 8"
for i in 1 2; do run VMV_TUNED_FILE=/tmp/old_table.json; run VMV_TUNED_FILE=/tmp/new_table.json; done
EXTRA="--simulate-rank 0 --latent 32x32"
for i in 1 2; do run VMV_TUNED_FILE=/tmp/old_table.json; run VMV_TUNED_FILE=/tmp/new_table.json; done
} 2>&1 | tee gpurun_out/r4m_ab.log

#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
( timeout 400 python tools/autotune_gemm.py --worlds "" --lgm --merge --out videomv_amd/tuned_gemm.json ) > gpurun_out/r4g_autotune_lgm.log 2>&1
cp videomv_amd/tuned_gemm.json gpurun_out/r4g_tuned_gemm.json
tail -n 3 gpurun_out/r4g_autotune_lgm.log
( timeout 300 python tools/experiments/tuned_parity.py ) > gpurun_out/r4g_tuned_parity.log 2>&1
grep -v amdgpu gpurun_out/r4g_tuned_parity.log | tail -8
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --simulate-rank 0 --no-i2vgen --no-op-profile"
( VMV_TUNED=0 timeout 300 $B ) > gpurun_out/r4g_bench_t0.json 2> gpurun_out/r4g_bench_t0.err
( VMV_TUNED=1 timeout 300 $B ) > gpurun_out/r4g_bench_t1.json 2> gpurun_out/r4g_bench_t1.err
for f in t0 t1; do python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/r4g_bench_$f.json') if l.startswith('{')][0])
    l=d['lgm_refined_sample']; s=d['sample_24view']
    print('$f', d['ms_per_step'], 'vae24', s['vae_decode24_seconds'], 'lgm step', l['lgm_refined_step_ms'], 'plain32', l['plain_step_ms'], 'ddim50_lgm', l['ddim50_lgm_seconds'], l['finite'], s['finite'])
except Exception as e:
    print('$f ERR', e); print(open('gpurun_out/r4g_bench_$f.err').read()[-600:])
PY
done

#!/bin/bash
# round 4, second GPU call: wide-tile LN / GEGLU epilogues (tests + A/B), GEMM autotune (worlds 1, 8, 4, 2), tuned-table A/B
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "layernorm_folded or geglu" ) > gpurun_out/r4b_t_kern.log 2>&1
( timeout 900 python -m pytest tests/test_unet_gpu.py -q -x -k "blocks_match or full_size_architecture or i2vgen_properties or reference_golden" ) > gpurun_out/r4b_t_unet.log 2>&1
B="python bench.py --steps 10 --warmup 3 --no-sample --no-cpu-baseline"
( VMV_TUNED=0 VMV_GEMM_XEPI=0 timeout 300 $B --simulate-rank 0 ) > gpurun_out/r4b_bench_xepi0.json 2> gpurun_out/r4b_bench_xepi0.err
( VMV_TUNED=0 VMV_GEMM_XEPI=1 timeout 300 $B --simulate-rank 0 --dump-ops gpurun_out/r4b_ops_xepi1.tsv ) > gpurun_out/r4b_bench_xepi1.json 2> gpurun_out/r4b_bench_xepi1.err
( timeout 900 python tools/autotune_gemm.py --worlds 1,8,4,2 --out videomv_amd/tuned_gemm.json ) > gpurun_out/r4b_autotune.log 2>&1
cp videomv_amd/tuned_gemm.json gpurun_out/r4b_tuned_gemm.json
( VMV_TUNED=0 timeout 400 $B ) > gpurun_out/r4b_bench_tuned0.json 2> gpurun_out/r4b_bench_tuned0.err
( VMV_TUNED=1 timeout 400 $B --dump-ops gpurun_out/r4b_ops_tuned1.tsv --dump-ops-sim gpurun_out/r4b_ops_sim_tuned1.tsv ) > gpurun_out/r4b_bench_tuned1.json 2> gpurun_out/r4b_bench_tuned1.err
for f in gpurun_out/r4b_t_kern.log gpurun_out/r4b_t_unet.log gpurun_out/r4b_autotune.log; do echo "== $f"; tail -n 4 $f; done
for f in xepi0 xepi1 tuned0 tuned1; do echo "== $f"; python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/r4b_bench_$f.json') if l.startswith('{')][0])
    s=d.get('simulated_rank') or {}
    print(d['ms_per_step'], d['roofline']['frac'], {k:v.get('gpu_ms_per_step') for k,v in (s.get('modes') or {}).items()})
except Exception as e:
    print('ERR', e); print(open('gpurun_out/r4b_bench_$f.err').read()[-800:])
PY
done

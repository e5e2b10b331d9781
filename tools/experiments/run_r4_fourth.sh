#!/bin/bash
# round 4, fourth GPU call: tile table for the 24x32x32 shapes (+ A/B), the frame-parallel legs through real RCCL at world 1, full GPU suite
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
B="python bench.py --steps 10 --warmup 3 --no-sample --no-cpu-baseline --simulate-rank 0"
( timeout 300 python tools/autotune_gemm.py --worlds 1 --latent 32x32 --merge --out videomv_amd/tuned_gemm.json ) > gpurun_out/r4f_autotune32.log 2>&1
cp videomv_amd/tuned_gemm.json gpurun_out/r4f_tuned_gemm.json
( VMV_TUNED=0 timeout 200 $B --latent 32x32 ) > gpurun_out/r4f_b32_t0.json 2> gpurun_out/r4f_b32_t0.err
( VMV_TUNED=1 timeout 200 $B --latent 32x32 ) > gpurun_out/r4f_b32_t1.json 2> gpurun_out/r4f_b32_t1.err
( VMV_BENCH_FORCE_PG=1 VMV_COMM_FORCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 400 $B ) > gpurun_out/r4f_fp_world1.json 2> gpurun_out/r4f_fp_world1.err
( VMV_COMM_NATIVE=0 VMV_BENCH_FORCE_PG=1 VMV_COMM_FORCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29534 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 400 $B --no-op-profile ) > gpurun_out/r4f_fp_world1_py.json 2> gpurun_out/r4f_fp_world1_py.err
tail -n 3 gpurun_out/r4f_autotune32.log
for f in b32_t0 b32_t1 fp_world1 fp_world1_py; do python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/r4f_$f.json') if l.startswith('{')][0])
    fp=d.get('frame_parallel') or {}
    print('$f', d['ms_per_step'], {k:(v.get('ms_per_step') if isinstance(v,dict) else v) for k,v in fp.items() if k in ('single_plan','branch_pipelined','branch_pipelined_graph','kv_gather_temporal','collectives_issued_by','error')})
except Exception as e:
    print('$f ERR', e); print(open('gpurun_out/r4f_$f.err').read()[-600:])
PY
done
( timeout 900 python -m pytest tests -m gpu -x -v --timeout 300 -p no:cacheprovider ) > gpurun_out/r4f_tests.log 2>&1
tail -n 4 gpurun_out/r4f_tests.log

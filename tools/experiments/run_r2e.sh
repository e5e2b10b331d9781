python -m pytest tests/test_frame_parallel_gpu.py -q 2>&1 | tail -8 > gpurun_out/r2e_pytest.log
cat gpurun_out/r2e_pytest.log
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-sample --no-lgm --no-op-profile"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
VMV_BENCH_FORCE_PG=1 VMV_COMM_FORCE=1 VMV_FP_PIPELINE=0 $B > gpurun_out/r2e_fp0.json 2> gpurun_out/r2e.err
VMV_BENCH_FORCE_PG=1 VMV_COMM_FORCE=1 VMV_FP_PIPELINE=1 $B > gpurun_out/r2e_fp1.json 2>> gpurun_out/r2e.err
tail -5 gpurun_out/r2e.err
python -c "
import json
for n in ('fp0','fp1'):
    d=json.load(open(f'gpurun_out/r2e_{n}.json')); print(n, d['value'], d['ms_per_step'], d['frame_parallel'])
"

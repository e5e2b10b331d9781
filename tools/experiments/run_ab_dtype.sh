set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r2_pytest_f16.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sample --no-lgm > gpurun_out/r2_bench_f16.json 2> gpurun_out/r2_bench_f16.err
VMV_DTYPE=bf16 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sample --no-lgm > gpurun_out/r2_bench_bf16.json 2> gpurun_out/r2_bench_bf16.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sample --no-lgm --no-op-profile > gpurun_out/r2_bench_f16_b.json 2>> gpurun_out/r2_bench_f16.err
VMV_DTYPE=bf16 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sample --no-lgm --no-op-profile > gpurun_out/r2_bench_bf16_b.json 2>> gpurun_out/r2_bench_bf16.err
cat gpurun_out/r2_pytest_f16.log
python -c "
import json
for n in ('f16','bf16','f16_b','bf16_b'):
    d=json.load(open(f'gpurun_out/r2_bench_{n}.json')); print(n, d['value'], d['ms_per_step'], d['finite'], d['dtype'])
"

#!/bin/bash
# GroupNorm after the replica records: kernel tests, the stats microbench, the frame-parallel tests, a bench step
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "groupnorm or gn_" 2>&1 | tail -3
timeout 120 python tools/experiments/gn_bench.py 2>&1 | grep "chunk_rows= 240\|chunk_rows= 102\|nchunk=  256\|nchunk=   13"
timeout 400 python -m pytest tests/test_frame_parallel_gpu.py tests/test_unet_gpu.py -x -q 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline --no-sample --steps 10 --warmup 2 > gpurun_out/gn_bench_step.json 2> gpurun_out/gn_bench_step.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/gn_bench_step.json").read().strip().splitlines()[-1])
print("STEP", d["ms_per_step"], {k: v["ms"] for k, v in d["roofline"]["families"].items()})
PY

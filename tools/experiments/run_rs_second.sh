#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "rs_" > $O/rs2_pytest.log 2>&1; echo "pytest rc=$?" >> $O/rs2_pytest.log
tail -4 $O/rs2_pytest.log
export VMV_BENCH_SHAPES="L0 N,L1 N"
timeout 600 python tools/gemm_bench.py 23 24 25 2>&1 | grep -v amdgpu.ids | tee $O/rs2_gemm_new.log
unset VMV_BENCH_SHAPES
P="--no-cpu-baseline --no-sample --no-op-profile --no-lgm --no-alt-dtype"
VMV_GEMM_RS=0 timeout 600 python bench.py $P --steps 15 --warmup 3 > $O/rs2_bench_off.json 2> $O/rs2_bench_off.err
timeout 600 python bench.py $P --steps 15 --warmup 3 > $O/rs2_bench_on.json 2> $O/rs2_bench_on.err
timeout 600 python bench.py $P --steps 15 --warmup 3 > $O/rs2_bench_on2.json 2> $O/rs2_bench_on2.err
for f in off on on2; do python - <<PY
import json
try:
    d=json.loads(open("$O/rs2_bench_$f.json").read().strip().splitlines()[-1]); print("$f", d["ms_per_step"], d["value"])
except Exception as e:
    print("$f", "failed", e)
PY
done
timeout 900 python -m pytest tests/test_unet_gpu.py -q -x 2>&1 | tail -4

#!/bin/bash
# first GPU contact of the row-stationary GEMM (gemm_rs.hip): parity tests, per-shape A/B against the previous policy, step A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "rs_" > $O/rs_pytest.log 2>&1; echo "pytest rc=$?" >> $O/rs_pytest.log
tail -5 $O/rs_pytest.log
export VMV_BENCH_SHAPES="L0 N,L1 N"
VMV_GEMM_RS=0 timeout 600 python tools/gemm_bench.py 0 > $O/rs_gemm_old.log 2>&1
timeout 600 python tools/gemm_bench.py 23 24 25 > $O/rs_gemm_new.log 2>&1
paste -d'|' $O/rs_gemm_old.log $O/rs_gemm_new.log | cut -c1-200
unset VMV_BENCH_SHAPES
P="--no-cpu-baseline --no-sample --no-op-profile --no-lgm --no-alt-dtype"
VMV_GEMM_RS=0 timeout 600 python bench.py $P --steps 15 --warmup 3 > $O/rs_bench_off.json 2> $O/rs_bench_off.err
timeout 600 python bench.py $P --steps 15 --warmup 3 > $O/rs_bench_on.json 2> $O/rs_bench_on.err
VMV_GEMM_RS=0 timeout 600 python bench.py $P --steps 15 --warmup 3 > $O/rs_bench_off2.json 2> $O/rs_bench_off2.err
timeout 600 python bench.py $P --steps 15 --warmup 3 > $O/rs_bench_on2.json 2> $O/rs_bench_on2.err
for f in off on off2 on2; do python - <<PY
import json
try:
    d=json.loads(open("$O/rs_bench_$f.json").read().strip().splitlines()[-1]); print("$f", d["ms_per_step"], d["value"])
except Exception as e:
    print("$f", "failed", e)
PY
done
tail -3 $O/rs_bench_on.err

#!/bin/bash
# full GPU suite + step A/B (row-stationary kernel on / off) + per-op table
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu -x > $O/full_pytest.log 2>&1; echo "pytest rc=$?" >> $O/full_pytest.log
tail -6 $O/full_pytest.log
P="--no-cpu-baseline --no-sample --no-op-profile --no-lgm --no-alt-dtype"
VMV_GEMM_RS=0 timeout 600 python bench.py $P --steps 15 --warmup 3 > $O/chk_bench_off.json 2> $O/chk_bench_off.err
timeout 600 python bench.py $P --steps 15 --warmup 3 > $O/chk_bench_on.json 2> $O/chk_bench_on.err
timeout 600 python bench.py --no-cpu-baseline --no-sample --no-lgm --no-alt-dtype --dump-ops $O/chk_ops_40x64.tsv > $O/chk_bench_ops.json 2> $O/chk_bench_ops.err
for f in off on ops; do python - <<PY
import json
try:
    d=json.loads(open("$O/chk_bench_$f.json").read().strip().splitlines()[-1]); print("$f", d["ms_per_step"], d["value"], d.get("roofline",{}).get("frac"))
except Exception as e:
    print("$f", "failed", e)
PY
done

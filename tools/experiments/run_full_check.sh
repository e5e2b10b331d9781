#!/bin/bash
# full GPU suite + bench (default flags minus the CPU baseline) + per-op table
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -q -m gpu > $O/full_pytest.log 2>&1; echo "pytest rc=$?" >> $O/full_pytest.log
tail -8 $O/full_pytest.log
timeout 900 python bench.py --no-cpu-baseline --no-lgm --no-alt-dtype --dump-ops $O/chk_ops_40x64.tsv > $O/chk_bench_ops.json 2> $O/chk_bench_ops.err
python - <<PY
import json
d=json.loads(open("$O/chk_bench_ops.json").read().strip().splitlines()[-1]); print("bench", d["ms_per_step"], d["value"], d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("families"))
PY

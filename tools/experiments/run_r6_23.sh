#!/bin/bash
# round 6, 23rd GPU call: the driver's round-end sequence on the final tree — smoke, the whole GPU tier, the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd $R; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/r6_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/r6_smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r6_gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -3 $O/r6_gpu_suite.log
S=$(date +%s); timeout 900 python bench.py > $O/r6_bench_head.json 2> $O/r6_bench_head.err; echo "bench rc=$? seconds=$(( $(date +%s) - S ))"
python -c "
import json
d=json.loads([l for l in open('$O/r6_bench_head.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic_stale'], d['reference_shape']['ms_per_step'])
print(json.dumps(d['prompt_batch'])[:700]); print(json.dumps(d['i2vgen']['shapes'])[:900]); print(json.dumps(d['sample_24view']))"

#!/bin/bash
# round 6, 17th GPU call: two prompts per plan — GPU parity test, then the bench leg at both shapes
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd $R; mkdir -p $O
timeout 900 python -m pytest tests/test_unet_gpu.py -x -q -k "two_prompts or fused_cfg_ddim_loop" > $O/r6_pbatch_tests.log 2>&1; tail -5 $O/r6_pbatch_tests.log
timeout 600 python bench.py --no-cpu-baseline --no-sample --simulate-rank 0 > $O/r6_pbatch_bench.json 2> $O/r6_pbatch_bench.err
python -c "
import json
d=json.loads([l for l in open('$O/r6_pbatch_bench.json') if l.startswith('{')][-1])
print(d['ms_per_step'], json.dumps(d['prompt_batch'], indent=1)); print(d['reference_shape'])"
tail -3 $O/r6_pbatch_bench.err

#!/bin/bash
# round 6, 25th GPU call: the LGM-refined 50-step loop with two prompts per plan (bench leg lgm_refined_sample.two_prompts_per_plan)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd $R; mkdir -p $O
timeout 900 python bench.py --no-cpu-baseline --no-op-profile --no-i2vgen --simulate-rank 0 > $O/r6_lgm_pbatch.json 2> $O/r6_lgm_pbatch.err
python -c "
import json
d=json.loads([l for l in open('$O/r6_lgm_pbatch.json') if l.startswith('{')][-1])
l=d['lgm_refined_sample']; print(d['ms_per_step'], {k:l[k] for k in ('ddim50_lgm_seconds','lgm_refined_step_ms','plain_step_ms')}, l.get('two_prompts_per_plan'))"
tail -2 $O/r6_lgm_pbatch.err

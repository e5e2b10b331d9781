#!/bin/bash
# round 6, 21st GPU call: b prompts per plan on the simulated 8-GPU rank (bench --simulate-rank 8)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd $R; mkdir -p $O
timeout 900 python bench.py --no-cpu-baseline --no-sample --no-op-profile --simulate-rank 8 > $O/r6_sim8_prompts.json 2> $O/r6_sim8_prompts.err
python -c "
import json
d=json.loads([l for l in open('$O/r6_sim8_prompts.json') if l.startswith('{')][-1])
s=d['simulated_rank']; print(d['ms_per_step'], {k:v.get('gpu_ms_per_step') for k,v in s['modes'].items()}); print(json.dumps(s.get('prompts_per_plan'), indent=1)); print(s.get('error'))"
tail -3 $O/r6_sim8_prompts.err

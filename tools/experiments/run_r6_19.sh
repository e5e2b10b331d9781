#!/bin/bash
# round 6, 19th GPU call: prompts-per-plan scaling (1..4) at both shapes, then the bench line with the two-prompt legs
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd $R; mkdir -p $O
{ timeout 500 python tools/experiments/prompt_batch_scaling.py 32x32 4; timeout 600 python tools/experiments/prompt_batch_scaling.py 40x64 4; } 2>/dev/null | tee $O/r6_prompt_batch_scaling.log
timeout 900 python bench.py --no-cpu-baseline --no-lgm --no-i2vgen --simulate-rank 0 > $O/r6_pbatch3_bench.json 2> $O/r6_pbatch3_bench.err
python -c "
import json
d=json.loads([l for l in open('$O/r6_pbatch3_bench.json') if l.startswith('{')][-1])
print(d['ms_per_step'], json.dumps(d['prompt_batch'])); print(json.dumps(d['sample_24view']))"
tail -3 $O/r6_pbatch3_bench.err

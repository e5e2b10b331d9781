#!/bin/bash
# round 6, 24th GPU call: per-family table of the two-prompt plan beside the one-prompt plan at 24x32x32 (bench leg prompt_batch)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd $R; mkdir -p $O
timeout 600 python bench.py --no-cpu-baseline --no-sample --simulate-rank 0 > $O/r6_pbatch_families.json 2> $O/r6_pbatch_families.err
python -c "
import json
d=json.loads([l for l in open('$O/r6_pbatch_families.json') if l.startswith('{')][-1])
q=d['prompt_batch']['24x32x32']['prompts_2']
print(d['ms_per_step'], q['ms_per_batched_step'])
for k in sorted(q['families_1_prompt']):
    a,b=q['families_1_prompt'][k], q['families'].get(k,{})
    print(f\"{k:10s} 1 prompt {a['ms']:7.3f} ms x2 = {2*a['ms']:7.3f} ({a['tflops']})   2 prompts {b.get('ms')} ms ({b.get('tflops')})  launches {a['launches']} / {b.get('launches')}\")"
tail -2 $O/r6_pbatch_families.err

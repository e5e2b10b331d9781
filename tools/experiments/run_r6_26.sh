#!/bin/bash
# round 6, 26th GPU call: rocprofv3 kernel-trace of the fused step at 24x32x32 with 1 and 2 prompts per plan (evidence behind DESIGN 4's family table)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_pb -- python $R/tools/experiments/prompt_batch_scaling.py 32x32 2 > $O/r6_prompts_32x32_kt.log 2>&1
cd $R
python tools/prof_summary.py $O/prof_pb $O/r6_kernel_stats_prompts_32x32.txt
rm -rf $O/prof_pb
head -14 $O/r6_kernel_stats_prompts_32x32.txt; grep "prompts/plan" $O/r6_prompts_32x32_kt.log

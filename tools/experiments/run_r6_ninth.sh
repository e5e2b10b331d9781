#!/bin/bash
# round 6, ninth GPU call: rasteriser with the pairwise packed blend (tests + bench), LGM-refined step gap analysis
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R; python -m pytest tests/test_gs_gpu.py -x -q > $O/r6_gs2_tests.log 2>&1; tail -2 $O/r6_gs2_tests.log
cd /tmp && export TMPDIR=/tmp
python $R/tools/experiments/gs_bench.py 5 > $O/r6_gs_bench_pair.log 2>&1; cat $O/r6_gs_bench_pair.log
VMV_GS_BATCH=1 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_gs -- python $R/tools/experiments/gs_bench.py 5 > $O/prof_gs.log 2>&1
cd $R; python tools/prof_summary.py $O/prof_gs $O/r6_gs_kernel_stats_pair.txt; rm -rf $O/prof_gs; grep "batch_kernel" $O/r6_gs_kernel_stats_pair.txt | head -8
cd /tmp
rocprofv3 --kernel-trace -f csv -d $O/prof_lgm -- python $R/tools/experiments/lgm_gaps.py run > $O/r6_lgm_gaps.log 2>&1
cd $R; python tools/experiments/lgm_gaps.py gaps $O/prof_lgm >> $O/r6_lgm_gaps.log 2>&1; rm -rf $O/prof_lgm; tail -32 $O/r6_lgm_gaps.log

R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -f csv -d $O/prof_lgm -- python $R/tools/experiments/lgm_gaps.py run > $O/r6_lgm_gaps.log 2>&1
cd $R; python tools/experiments/lgm_gaps.py gaps $O/prof_lgm >> $O/r6_lgm_gaps.log 2>&1; rm -rf $O/prof_lgm; grep -v "rocprofv3\|output_stream\|tool.cpp" $O/r6_lgm_gaps.log | tail -40

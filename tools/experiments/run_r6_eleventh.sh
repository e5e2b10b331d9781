#!/bin/bash
# round 6, eleventh GPU call: LGM-refined step after the async posterior-noise upload + cached camera constants
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd $R; python -m pytest tests -m gpu -x -q -k "lgm or gs or vae or entrance or posterior" > $O/r6_lgm_tests.log 2>&1; tail -2 $O/r6_lgm_tests.log
cd /tmp && export TMPDIR=/tmp
python $R/tools/experiments/lgm_gaps.py run > $O/r6_lgm_steps_noprof.log 2>&1; grep "wall" $O/r6_lgm_steps_noprof.log
rocprofv3 --kernel-trace -f csv -d $O/prof_lgm -- python $R/tools/experiments/lgm_gaps.py run > $O/r6_lgm_gaps_after.log 2>&1
cd $R; python tools/experiments/lgm_gaps.py gaps $O/prof_lgm >> $O/r6_lgm_gaps_after.log 2>&1; rm -rf $O/prof_lgm; grep -v "rocprofv3\|output_stream\|tool.cpp" $O/r6_lgm_gaps_after.log | tail -16

#!/bin/bash
# round 6, fourth GPU call: the whole GPU suite on the final kernels (incl. bench.py --gpus 2 end to end on one GPU), step A/B of the
# round's three switches against everything-off, then the round's profile bundle
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests -x -q -m gpu > $O/r6_gpu_suite.log 2>&1; echo "suite rc $?" >> $O/r6_gpu_suite.log; tail -4 $O/r6_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/r6_smoke.log 2>&1; tail -2 $O/r6_smoke.log
bash tools/experiments/run_env_ab2.sh "VMV_TQA=0 VMV_XGLDS_GM=1 VMV_GLDS_GM=1" "VMV_XGLDS_GM=-1" > $O/r6_round_step_ab.log 2>&1; cat $O/r6_round_step_ab.log
RN=r6 bash tools/profile_round.sh > $O/r6_bundle.log 2>&1; tail -5 $O/r6_bundle.log

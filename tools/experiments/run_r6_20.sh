#!/bin/bash
# round 6, 20th GPU call: tile table for the two- / four-prompt plans (M doubled / quadrupled), A/B of the batched step with the old / new table
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd $R; mkdir -p $O
cp videomv_amd/tuned_gemm.json $O/r6_pb_tuned_gemm.json
for spec in "32x32 2" "32x32 4" "40x64 2"; do set -- $spec
  timeout 600 python tools/autotune_gemm.py --worlds 1 --latent $1 --prompts $2 --out $O/r6_pb_tuned_gemm.json --merge > $O/r6_pb_tune_$1_p$2.log 2>&1; tail -2 $O/r6_pb_tune_$1_p$2.log
done
for v in old new old new; do
  if [ $v = new ]; then export VMV_TUNED_FILE=$O/r6_pb_tuned_gemm.json; else unset VMV_TUNED_FILE; fi
  echo "== table $v"; timeout 400 python tools/experiments/prompt_batch_scaling.py 32x32 4 2>/dev/null | grep -v "plan 3"
  timeout 400 python tools/experiments/prompt_batch_scaling.py 40x64 2 2>/dev/null
done | tee $O/r6_pb_tune_ab.log

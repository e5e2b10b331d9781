#!/bin/bash
# round 6, 14th GPU call: the tile table was measured in round 4; the kernels it chooses between have changed since (grouped tile order,
# run-wise K walk, K = 512 row-stationary, 8 x 1 wave grid).  Re-tune the world-1 signatures at 40x64 and 32x32 into a COPY of the
# table and A/B the step with the old / new file on the same box.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd $R; mkdir -p $O
cp videomv_amd/tuned_gemm.json $O/r6_retuned_gemm.json
timeout 600 python tools/autotune_gemm.py --worlds 1 --latent 40x64 --out $O/r6_retuned_gemm.json --merge > $O/r6_retune_40x64.log 2>&1; tail -3 $O/r6_retune_40x64.log
timeout 600 python tools/autotune_gemm.py --worlds 1 --latent 32x32 --out $O/r6_retuned_gemm.json --merge > $O/r6_retune_32x32.log 2>&1; tail -3 $O/r6_retune_32x32.log
for lat in 40x64 32x32; do
for v in old new old new; do
  if [ $v = new ]; then export VMV_TUNED_FILE=$O/r6_retuned_gemm.json; else unset VMV_TUNED_FILE; fi
  timeout 200 python bench.py --latent $lat --no-cpu-baseline --no-sample --no-op-profile --no-lgm --no-i2vgen --simulate-rank 0 --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('STEP $lat $v', d['ms_per_step'])"
done; done 2>&1 | tee $O/r6_retune_step_ab.log

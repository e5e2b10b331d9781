#!/bin/bash
# round 6, fifth GPU call: 64-query attention blocks for the short problems + two-level total fold in the one-launch GroupNorm
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests/test_kernels_gpu.py -x -q -k "attention or groupnorm" > $O/r6_attn_gn_tests.log 2>&1; echo "tests rc $?" >> $O/r6_attn_gn_tests.log; tail -3 $O/r6_attn_gn_tests.log
for v in 0 1 0 1; do
  VMV_ATTN_Q64=$v python bench.py --no-cpu-baseline --no-sample --steps 10 --warmup 2 --simulate-rank 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['roofline']['families']; print('STEP VMV_ATTN_Q64=$v', d['ms_per_step'], 'attention', f['attention']['ms'], 'gn_fused', f['gn_fused']['ms'], '32x32', d['reference_shape']['ms_per_step'])"
done > $O/r6_attn_q64_step_ab.log 2>&1; cat $O/r6_attn_q64_step_ab.log
python tools/experiments/gnf_stamps.py > $O/r6_gnf_stamps_fold.log 2>&1; cat $O/r6_gnf_stamps_fold.log
python -m pytest tests/test_unet_gpu.py -x -q -k "golden or full_size_reference or tiny or block" > $O/r6_attn_unet_tests.log 2>&1; tail -2 $O/r6_attn_unet_tests.log

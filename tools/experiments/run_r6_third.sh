#!/bin/bash
# round 6, third GPU call: one-launch GroupNorm with LDS-DMA staging — parity tests, kernel-trace durations, same-box step A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests/test_kernels_gpu.py -x -q -k "groupnorm or gemm_rs" > $O/r6_gn_tests.log 2>&1; echo "tests rc $?" >> $O/r6_gn_tests.log; tail -3 $O/r6_gn_tests.log
bash tools/experiments/run_env_ab.sh VMV_GNF_DMA 0 1 > $O/r6_gnf_dma_step_ab.log 2>&1; cat $O/r6_gnf_dma_step_ab.log
for v in 0 1; do
  VMV_GNF_DMA=$v python bench.py --latent 32x32 --no-cpu-baseline --no-sample --no-op-profile --simulate-rank 0 --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('32x32 VMV_GNF_DMA=$v', d['ms_per_step'])"
done > $O/r6_gnf_dma_32x32.log 2>&1; cat $O/r6_gnf_dma_32x32.log
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  VMV_GNF_DMA=$v rocprofv3 --kernel-trace --stats -f csv -d $O/kt_gnf$v -- python $R/bench.py --no-cpu-baseline --no-sample --no-op-profile --simulate-rank 0 --steps 3 --warmup 1 > $O/kt_gnf$v.log 2>&1
  python $R/tools/prof_summary.py $O/kt_gnf$v $O/r6_gnf_dma${v}_kernel_stats.txt; grep -m1 "gn_fused" $O/r6_gnf_dma${v}_kernel_stats.txt
  rm -rf $O/kt_gnf$v
done
cd $R
python -m pytest tests/test_unet_gpu.py -x -q -k "golden or full_size_reference or tiny" > $O/r6_gn_unet_tests.log 2>&1; tail -2 $O/r6_gn_unet_tests.log
python tools/experiments/rs512_bench.py > $O/r6_rs512_bench.log 2>&1; cat $O/r6_rs512_bench.log
for e in "VMV_XGLDS_GM=1 VMV_GLDS_GM=1" "VMV_XGLDS_GM=-1"; do env $e python tools/experiments/groupm_bench.py; done > $O/r6_groupm_bench.log 2>&1; cat $O/r6_groupm_bench.log
bash tools/experiments/run_env_ab2.sh "VMV_XGLDS_GM=1 VMV_GLDS_GM=1" "VMV_XGLDS_GM=-1" > $O/r6_groupm_step_ab.log 2>&1; cat $O/r6_groupm_step_ab.log

#!/bin/bash
# round 6, seventh GPU call: run-wise K walk in gemm_glds too — kernel tests, conv microbench A/B, step A/B, per-kernel traffic table
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm or conv" > $O/r6_tap2_tests.log 2>&1; echo "tests rc $?" >> $O/r6_tap2_tests.log; tail -3 $O/r6_tap2_tests.log
for v in 0 1; do
  echo "# VMV_GLDS_TAPMAJOR=$v"
  VMV_GLDS_TAPMAJOR=$v VMV_BENCH_SHAPES="conv L2,conv L3,tcnv L3,tcnv L1" python tools/gemm_bench.py 0 6 5 2>/dev/null
done > $O/r6_tap2_bench.log 2>&1; cat $O/r6_tap2_bench.log
bash tools/experiments/run_env_ab2.sh "VMV_GLDS_TAPMAJOR=0 VMV_XGLDS_TAPMAJOR=0" "VMV_GLDS_TAPMAJOR=1" > $O/r6_tap2_step_ab.log 2>&1; cat $O/r6_tap2_step_ab.log
python -m pytest tests/test_unet_gpu.py -x -q -k "golden or full_size_reference or tiny or block" > $O/r6_tap2_unet_tests.log 2>&1; tail -2 $O/r6_tap2_unet_tests.log
RN=r6b
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -f csv -d $O/tb_fetch -- python $R/tools/traffic_by_op.py run $O/${RN}_traffic_ops.json > $O/tb_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -f csv -d $O/tb_write -- python $R/tools/traffic_by_op.py run $O/${RN}_traffic_ops.json > $O/tb_write.log 2>&1
cd $R
python tools/traffic_by_op.py table $O/${RN}_traffic_ops.json $O/tb_fetch $O/tb_write $O/${RN}_gemm_traffic_by_kernel.tsv $O/${RN}_gemm_traffic.json
rm -rf $O/tb_fetch $O/tb_write
head -3 $O/${RN}_gemm_traffic_by_kernel.tsv

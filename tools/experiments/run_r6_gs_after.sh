cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd $R; python -m pytest tests/test_gs_gpu.py -x -q > $O/r6_gs_tests.log 2>&1; tail -3 $O/r6_gs_tests.log
cd /tmp
python $R/tools/experiments/gs_bench.py 5 > $O/r6_gs_bench_after.log 2>&1
GS_SMAX=0.12 python $R/tools/experiments/gs_bench.py 3 >> $O/r6_gs_bench_after.log 2>&1
VMV_GS_BATCH=1 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_gs -- python $R/tools/experiments/gs_bench.py 5 > $O/prof_gs.log 2>&1
cd $R; python tools/prof_summary.py $O/prof_gs $O/r6_gs_kernel_stats_after.txt; rm -rf $O/prof_gs
cat $O/r6_gs_bench_after.log; grep -i "batch\|rocprim" $O/r6_gs_kernel_stats_after.txt | cut -c1-200 | head -30

cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
python $R/tools/experiments/gs_bench.py 5 > $O/r6_gs_bench_before.log 2>&1
VMV_GS_BATCH=1 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_gs -- python $R/tools/experiments/gs_bench.py 5 > $O/prof_gs.log 2>&1
cd $R; python tools/prof_summary.py $O/prof_gs $O/r6_gs_kernel_stats_before.txt; rm -rf $O/prof_gs
cat $O/r6_gs_bench_before.log; head -30 $O/r6_gs_kernel_stats_before.txt

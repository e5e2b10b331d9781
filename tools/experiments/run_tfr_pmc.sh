# rocprofv3 counters of the frame-resident temporal convolution beside the 256 x 320 tile kernel on the first level's shapes
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export TFR_BENCH_REPS=3 TFR_BENCH_ONLY="L0 "
rocprofv3 --kernel-trace --stats -f csv -d $O/tfr_kt -- python $R/tools/experiments/tfr_bench.py > $O/tfr_kt.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU -f csv -d $O/tfr_pmc -- python $R/tools/experiments/tfr_bench.py > $O/tfr_pmc.log 2>&1
rocprofv3 --pmc FETCH_SIZE -f csv -d $O/tfr_fetch -- python $R/tools/experiments/tfr_bench.py > $O/tfr_fetch.log 2>&1
cd $R
python tools/prof_summary.py $O/tfr_kt $O/r5_tfr_kernel_stats.txt
python tools/prof_summary.py $O/tfr_pmc $O/r5_tfr_pmc.txt
python tools/prof_summary.py $O/tfr_fetch $O/r5_tfr_fetch.txt
grep -E "gemm_tfr|gemm_xglds|gn_apply" $O/r5_tfr_kernel_stats.txt | head; grep -E "gemm_tfr|gemm_xglds" $O/r5_tfr_pmc.txt | head -30
rm -rf $O/tfr_kt $O/tfr_pmc $O/tfr_fetch

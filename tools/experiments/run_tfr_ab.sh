set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "tfr" > $O/b_tfr_tests.log 2>&1; echo "tfr tests rc=$?" | tee -a $O/b_tfr_tests.log
tail -4 $O/b_tfr_tests.log
timeout 300 python tools/experiments/tfr_bench.py > $O/b_tfr_bench.log 2>&1; echo "bench rc=$?"; cat $O/b_tfr_bench.log | tail -9

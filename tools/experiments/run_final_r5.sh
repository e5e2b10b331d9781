# final evidence of round 5: smoke, the whole GPU tier, then the profile bundle (tools/profile_round.sh)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > $O/r5_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/r5_smoke.log
python -m pytest tests -m gpu -x -q > $O/r5_gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -4 $O/r5_gpu_suite.log
RN=r5 bash tools/profile_round.sh > $O/r5_bundle.log 2>&1; echo "bundle rc=$?"; tail -c 400 $O/r5_bench_40x64.json

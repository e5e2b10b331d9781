# final evidence of round 6: smoke, the whole GPU tier, the vendor yardstick, then the profile bundle (tools/profile_round.sh)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > $O/r6_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/r6_smoke.log
python -m pytest tests -m gpu -x -q > $O/r6_gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -4 $O/r6_gpu_suite.log
python tools/vendor_yardstick.py --out $O/r6_vendor_yardstick.tsv > $O/r6_vendor_yardstick.log 2>&1; echo "yardstick rc=$?"; tail -3 $O/r6_vendor_yardstick.log
RN=r6 bash tools/profile_round.sh > $O/r6_bundle.log 2>&1; echo "bundle rc=$?"; tail -c 400 $O/r6_bench_40x64.json

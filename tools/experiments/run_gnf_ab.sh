# same-box A/B of the one-launch GroupNorm's loads in flight (4 = before, 8, 16): library builds in ab_libs/, picked with VMV_LIB_DIR
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "groupnorm" > $O/e_gn_tests.log 2>&1; echo "gn tests rc=$?"; tail -2 $O/e_gn_tests.log
B="python bench.py --no-cpu-baseline --no-sample --no-lgm --no-i2vgen --simulate-rank 0 --steps 15 --warmup 3"
for rep in 1 2; do
for v in before unr8 unr16; do
  VMV_LIB_DIR=$GRAFT_REPO_ROOT/ab_libs/$v timeout 300 $B 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
f = d['roofline']['families']
print('$v', d['ms_per_step'], 'gn_fused', f['gn_fused']['ms'], 'ref32', (d.get('reference_shape') or {}).get('ms_per_step'))" | tee -a $O/r5_gnf_unroll_ab.log
done; done

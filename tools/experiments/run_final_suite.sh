set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests -m gpu -q > $O/r5_gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -3 $O/r5_gpu_suite.log
B="python bench.py --no-cpu-baseline --no-sample --no-lgm --no-i2vgen --simulate-rank 0 --no-op-profile --steps 15 --warmup 3 --latent 48x48"
for cfg in "VMV_TUNED=0 VMV_TILE_RULES=0" "VMV_TUNED=0" "VMV_TUNED=1"; do
  env $cfg $B 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('24x48x48', '$cfg', d['ms_per_step'], d['finite'])" | tee -a $O/r5_48x48_rule_ab.log
done

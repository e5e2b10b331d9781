set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "wide_tile or temporal_conv or conv3x3 or gemm_basic or xglds" > $O/c_xglds_tests.log 2>&1; echo "xglds tests rc=$?"; tail -3 $O/c_xglds_tests.log
B="python bench.py --no-cpu-baseline --no-sample --no-lgm --no-i2vgen --simulate-rank 0 --steps 15 --warmup 3"
for rep in 1 2; do
for cfg in "VMV_LIB_DIR=$GRAFT_REPO_ROOT/ab_libs/before" "VMV_X=1"; do
  env $cfg timeout 300 $B --dump-ops $O/c_ops_$(echo $cfg | tr ' =/' '___' | tail -c 20).tsv 2> $O/c_step.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
f = d['roofline']['families']
print('$cfg'[-24:], d['ms_per_step'], 'gemm', f['gemm']['ms'], f['gemm']['tflops'], 'ref32', (d.get('reference_shape') or {}).get('ms_per_step'))"
done; done
python - <<'P'
import csv
for f in ("gpurun_out/c_ops__ab_libs_before.tsv","gpurun_out/c_ops_VMV_X_1.tsv"):
    try:
        rows=list(csv.DictReader(open(f),delimiter='\t'))
    except Exception as e:
        print(f, e); continue
    sel=[r for r in rows if r['label'].endswith('temopral_conv.conv4') or r['label'].endswith('.conv2')]
    print(f, round(sum(float(r['ms']) for r in sel),3), 'ms over', len(sel), 'residual convs;', [ (r['label'][-28:], r['ms']) for r in sel if 'output_blocks.11' in r['label'] or 'input_blocks.2.0' in r['label']])
P

set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r2b_pytest.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sample --no-lgm --no-op-profile"
VMV_ATTN_SHORT=0 VMV_SHARE_PREFIX=0 $B > gpurun_out/r2b_base.json 2> gpurun_out/r2b.err
VMV_ATTN_SHORT=1 VMV_SHARE_PREFIX=0 $B > gpurun_out/r2b_short.json 2>> gpurun_out/r2b.err
VMV_ATTN_SHORT=1 VMV_SHARE_PREFIX=1 $B > gpurun_out/r2b_short_share.json 2>> gpurun_out/r2b.err
VMV_ATTN_SHORT=0 VMV_SHARE_PREFIX=0 $B > gpurun_out/r2b_base2.json 2>> gpurun_out/r2b.err
python bench.py --steps 10 --warmup 3 --no-sample --no-lgm --dump-ops gpurun_out/r2b_ops.tsv > gpurun_out/r2b_full.json 2>> gpurun_out/r2b.err
cat gpurun_out/r2b_pytest.log
python -c "
import json
for n in ('base','short','short_share','base2','full'):
    d=json.load(open(f'gpurun_out/r2b_{n}.json')); print(n, d['value'], d['ms_per_step'], d['finite'], d['dtype'], d.get('cpu_baseline'))
"

#!/bin/bash
# round 6, second GPU call: the whole GPU suite, the per-op HBM traffic attribution (two PMC passes), the bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests -x -q -m gpu > $O/r6_gpu_suite_mid.log 2>&1; echo "suite rc $?" >> $O/r6_gpu_suite_mid.log
tail -4 $O/r6_gpu_suite_mid.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -f csv -d $O/tb_fetch -- python $R/tools/traffic_by_op.py run $O/r6_traffic_ops.json > $O/tb_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -f csv -d $O/tb_write -- python $R/tools/traffic_by_op.py run $O/r6_traffic_ops.json > $O/tb_write.log 2>&1
cd $R
python tools/traffic_by_op.py table $O/r6_traffic_ops.json $O/tb_fetch $O/tb_write $O/r6_gemm_traffic_by_kernel.tsv $O/r6_gemm_traffic.json > $O/tb_table.log 2>&1
cat $O/tb_table.log; head -30 $O/r6_gemm_traffic_by_kernel.tsv
rm -rf $O/tb_fetch $O/tb_write
cd /tmp
python $R/bench.py --dump-ops $O/r6_ops_40x64_mid.tsv > $O/r6_bench_mid.json 2> $O/r6_bench_mid.err
python -c "
import json; d=json.load(open('$O/r6_bench_mid.json'))
print('ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'whole', d['roofline']['whole_step']['frac'])
print('families', {k:(v['ms'],v['launches']) for k,v in d['roofline']['families'].items()})
print('ref_shape', d['reference_shape'])
print('sim', d['simulated_rank'] and {k:v.get('gpu_ms_per_step') for k,v in d['simulated_rank']['modes'].items()})
print('lgm', d['lgm_refined_sample'] and d['lgm_refined_sample'].get('lgm_refined_step_ms'), d['lgm_refined_sample'] and d['lgm_refined_sample'].get('rasteriser',{}).get('ms_per_24_views'))
print('i2v', d['i2vgen'] and {k:v['ms_per_step'] for k,v in d['i2vgen']['shapes'].items()})
print('cpu', d['cpu_baseline'] and d['cpu_baseline']['value'])
"

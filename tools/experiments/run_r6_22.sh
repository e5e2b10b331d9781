#!/bin/bash
# round 6, 22nd GPU call: UNetSD_I2VGen with two input images per plan — GPU parity test, bench leg
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd $R; mkdir -p $O
timeout 900 python -m pytest tests/test_unet_gpu.py -x -q -k "i2vgen" > $O/r6_i2v_batch_tests.log 2>&1; tail -5 $O/r6_i2v_batch_tests.log
timeout 900 python bench.py --no-cpu-baseline --no-lgm --no-op-profile --simulate-rank 0 > $O/r6_i2v_batch_bench.json 2> $O/r6_i2v_batch_bench.err
python -c "
import json
d=json.loads([l for l in open('$O/r6_i2v_batch_bench.json') if l.startswith('{')][-1])
print(d['ms_per_step'], json.dumps(d['i2vgen']['shapes'], indent=1))"
tail -3 $O/r6_i2v_batch_bench.err

#!/bin/bash
# round 4, first GPU call: probe + the new tests + bench (new legs) + vendor yardstick + slab-chain experiment
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 60 tools/experiments/f16_ovfl_probe ) > gpurun_out/r4_probe.log 2>&1
( timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "saturate" ) > gpurun_out/r4_t_sat.log 2>&1
( timeout 900 python -m pytest tests/test_frame_parallel_gpu.py -q -s ) > gpurun_out/r4_t_fp.log 2>&1
( timeout 1200 python -m pytest tests/test_unet_gpu.py -q -s -k "ddim50_full or i2vgen_properties or config1_properties" ) > gpurun_out/r4_t_unet.log 2>&1
( timeout 900 python bench.py --steps 10 --warmup 3 --dump-ops gpurun_out/r4_ops_40x64.tsv --dump-ops-sim gpurun_out/r4_ops_sim_rank0of8.tsv ) > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err
( timeout 600 python tools/vendor_yardstick.py --out gpurun_out/r4_vendor_yardstick.tsv ) > gpurun_out/r4_vendor.log 2>&1
( timeout 300 python tools/experiments/slab_chain.py ) > gpurun_out/r4_slab.log 2>&1
tail -3 gpurun_out/r4_probe.log gpurun_out/r4_t_sat.log gpurun_out/r4_t_fp.log gpurun_out/r4_t_unet.log gpurun_out/r4_slab.log
tail -c 1500 gpurun_out/r4_bench.err

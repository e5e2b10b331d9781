#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
RN=r4 VMV_COMMIT=$(cat .commit 2>/dev/null) timeout 900 bash tools/profile_round.sh > gpurun_out/r4_profile.log 2>&1
tail -c 300 gpurun_out/r4_profile.log
( timeout 300 python -m pytest tests/test_unet_gpu.py tests/test_kernels_gpu.py -m gpu -q -x --timeout 200 -p no:cacheprovider -k "reference_golden or full_size_config1 or full_size_forwards or saturate or fused_cfg" ) 2>&1 | tail -3

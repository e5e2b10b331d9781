#!/bin/bash
# round 4, third GPU call: the tests the -x run did not reach, then the round's profile bundle
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_unet_gpu.py -m gpu -v -s --timeout 300 -p no:cacheprovider -k "non_finite or inference_py or i2vgen or full_size or lgm or vae_encode or ddim50" ) > gpurun_out/r4e_tests.log 2>&1
grep -i "rel-L2\|PSNR\|passed\|failed\|PASSED\|FAILED" gpurun_out/r4e_tests.log | tail -30
RN=r4 VMV_COMMIT=$(cat .commit 2>/dev/null) timeout 1100 bash tools/profile_round.sh > gpurun_out/r4_profile.log 2>&1
tail -c 400 gpurun_out/r4_profile.log

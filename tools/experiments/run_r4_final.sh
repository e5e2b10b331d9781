#!/bin/bash
# round 4, final GPU call: full GPU suite at HEAD, then the profile bundle
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 800 python -m pytest tests -m gpu -x -v --timeout 300 -p no:cacheprovider ) > gpurun_out/r4h_tests.log 2>&1
tail -n 3 gpurun_out/r4h_tests.log
RN=r4 VMV_COMMIT=$(cat .commit 2>/dev/null) timeout 900 bash tools/profile_round.sh > gpurun_out/r4_profile.log 2>&1
tail -c 300 gpurun_out/r4_profile.log

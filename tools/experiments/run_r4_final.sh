#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
RN=r4 VMV_COMMIT=$(cat .commit 2>/dev/null) timeout 600 bash tools/profile_round.sh > gpurun_out/r4_profile.log 2>&1
tail -c 200 gpurun_out/r4_profile.log

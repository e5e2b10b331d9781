#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
( timeout 200 python tools/experiments/nan_probe.py ) > gpurun_out/r4_nan_probe.log 2>&1
cat gpurun_out/r4_nan_probe.log | grep -v amdgpu.ids

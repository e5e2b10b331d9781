#!/bin/bash
# experiment: GroupNorm statistics chunking (bytes per block, chunks per stat group) judged by the whole step; then the final bundle
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
B="python bench.py --steps 20 --warmup 5 --no-sample --no-cpu-baseline --simulate-rank 0 --no-op-profile"
run() { ( env "$@" timeout 120 $B ) 2>/dev/null | python -c "import sys,json; print('$*', json.loads([l for l in sys.stdin if l.startswith('{')][0])['ms_per_step'])"; }
for i in 1 2; do
  run VMV_GN_CHUNK_KB=64 VMV_GN_MAXCHUNKS=256
  run VMV_GN_CHUNK_KB=32 VMV_GN_MAXCHUNKS=512
  run VMV_GN_CHUNK_KB=32 VMV_GN_MAXCHUNKS=256
  run VMV_GN_CHUNK_KB=128 VMV_GN_MAXCHUNKS=256
  run VMV_GN_CHUNK_KB=64 VMV_GN_MAXCHUNKS=512
done 2>&1 | tee gpurun_out/r4j_gnchunk.log

#!/bin/bash
# round 6, 15th GPU call: the N > 1 bench flow with the frame-parallel leg in child processes, on ONE GPU (both ranks on device 0, gloo,
# host-staged collectives: a smoke test of the flow, never a measurement) — self-spawned form (the GPU test) and the driver's
# torch.distributed.run form.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd $R; mkdir -p $O
timeout 1500 python -m pytest tests/test_frame_parallel_gpu.py -x -q -k "bench_gpus_2" > $O/r6_fpchild_test.log 2>&1; tail -3 $O/r6_fpchild_test.log
VMV_BENCH_PG_BACKEND=gloo VMV_BENCH_SHARE_GPU=1 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --steps 2 --warmup 1 --latent 16x16 --frames 4 --no-cpu-baseline --no-op-profile --frame-parallel-budget 600 > $O/r6_fpchild_torchrun.json 2> $O/r6_fpchild_torchrun.err
echo "torchrun rc=$?"; python - <<PY
import json
l=[x for x in open("$O/r6_fpchild_torchrun.json") if x.startswith("{")]
print("json lines", len(l))
d=json.loads(l[-1]); fp=d.get("frame_parallel") or {}
print({k:d.get(k) for k in ("n_gpus","rccl_ranks","launcher","value","ms_per_step","finite")})
print({k:(v if not isinstance(v,dict) else {kk:v.get(kk) for kk in ("ms_per_step","finite","error")}) for k,v in fp.items()})
PY
tail -5 $O/r6_fpchild_torchrun.err

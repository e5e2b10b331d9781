#!/bin/bash
# round 6, 27th GPU call: the N = 4 and N = 8 bench flows END TO END on ONE GPU (all ranks on device 0, gloo, host-staged collectives — a
# smoke test of the world-4 / world-8 code paths on real kernels: replica timing, children, all seven frame-parallel sub-legs; never a measurement)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd $R; mkdir -p $O
free -g | head -2; nproc
for n in ${NS:-4 8}; do
  S=$(date +%s)
  VMV_BENCH_PG_BACKEND=gloo VMV_BENCH_SHARE_GPU=1 timeout 1500 python bench.py --gpus $n --steps 2 --warmup 1 --latent ${LAT:-16x16} --frames 24 --no-cpu-baseline --no-op-profile --frame-parallel-budget 900 > $O/r6_share_gpu_n$n.json 2> $O/r6_share_gpu_n$n.err
  echo "n=$n rc=$? seconds=$(( $(date +%s) - S ))"
  python - <<PY
import json
l=[x for x in open("$O/r6_share_gpu_n$n.json") if x.startswith("{")]
print("json lines", len(l))
if l:
    d=json.loads(l[-1]); fp=d.get("frame_parallel") or {}
    print({k:d.get(k) for k in ("n_gpus","rccl_ranks","launcher","value","ms_per_step","finite")})
    for k,v in fp.items():
        if isinstance(v,dict): print("  ",k,{kk:v.get(kk) for kk in ("ms_per_step","finite","error","2","4") if kk in v})
        elif k in ("error","mode","views_per_gpu","child_rccl_ranks","collectives_per_branch_plan","all_to_all_per_step"): print("  ",k,v)
PY
  grep -v "amdgpu.ids\|socket.cpp\|^$" $O/r6_share_gpu_n$n.err | tail -4
done

# step-level A/B of forced GEMM variants for the short-K linears (one box)
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sample --no-lgm --no-op-profile --no-alt-dtype"
run() { env "$@" $B 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$*', d['value'], d['ms_per_step'], d['finite'])"; }
run X=0
for t in 5 9 13 15 19; do run VMV_GEMM_TILE_GEGLU=$t; done
run X=0
for t in 6 10 14 16 18; do run VMV_GEMM_TILE_LIN160=$t; done
run X=0

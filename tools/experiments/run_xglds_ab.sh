# step-level A/B of the wide-tile conv kernel (gemm_xglds.hip): VMV_GEMM_XGLDS=0 / 1
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sample --no-lgm --no-op-profile --no-alt-dtype"
run() { env "$@" $B 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$*', d['value'], d['ms_per_step'], d['finite'])"; }
for i in 1 2 3; do
run VMV_GEMM_XGLDS=0
run VMV_GEMM_XGLDS=1
done

# step-level A/B of the in-loop LayerNorm statistics (VMV_LN_INLINE=1; default 0) against the separate statistics pass
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sample --no-lgm --no-op-profile --no-alt-dtype"
run() { env "$@" $B 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$*', d['value'], d['ms_per_step'], d['finite'])"; }
for i in 1 2 3; do
run VMV_LN_INLINE=0
run VMV_LN_INLINE=1
done
for v in 0 1; do echo "VMV_BENCH_LN_INLINE=$v"; VMV_BENCH_LN_INLINE=$v VMV_BENCH_SHAPES="lnqkv,lngeglu" python tools/gemm_bench.py 0; done

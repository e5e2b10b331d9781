# step-level A/B of the GEMM policies on one box (2 = default, 3 = wave-specialised kernel for every eligible GEMM, 4 = for gathers only)
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sample --no-lgm --no-op-profile"
for pol in 2 3 4 2; do VMV_GEMM_POLICY=$pol $B 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('policy $pol', d['value'], d['ms_per_step'])"; done

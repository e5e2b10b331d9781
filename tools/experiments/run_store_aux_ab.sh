# step-level A/B of write-through (sc1) output stores: builds of the same source with -DVMV_WT_MB / -DVMV_ATTN_WT_MB thresholds
# (csrc/Makefile EXTRA=...): never / default (32 MB) / always / attention only / GEMM only
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sample --no-lgm --no-op-profile --no-alt-dtype"
run() { env "$@" $B 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$*', d['value'], d['ms_per_step'], d['finite'])"; }
for i in 1 2; do
run X=default32
for v in never always attnonly gemmonly; do run VMV_LIB_DIR=$PWD/ab_libs/$v; done
done

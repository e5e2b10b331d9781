# schedule variants of gemm_xglds.hip (VMV_XGLDS_VARIANT builds in ab_libs/xv<k>): 1 DMA issue before the last MFMA phase,
# 2 s_setprio 1 around the MFMA phases, 3 both
#   for k in 1 2 3; do make -C videomv_amd/csrc EXTRA=-DVMV_XGLDS_VARIANT=$k LIBDIR=../../ab_libs/xv$k BUILD=build_xv$k ../../ab_libs/xv$k/libvmv_hip_f16.so; done
S="conv L0,tcnv L0,conv L1,tcnv L1"
echo "== product"; VMV_BENCH_SHAPES="$S" python tools/gemm_bench.py 20
for k in 1 2 3; do echo "== variant $k"; VMV_LIB_DIR=$PWD/ab_libs/xv$k VMV_BENCH_SHAPES="$S" python tools/gemm_bench.py 20; done
echo "== product again"; VMV_BENCH_SHAPES="$S" python tools/gemm_bench.py 20

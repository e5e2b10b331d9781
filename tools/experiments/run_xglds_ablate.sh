# ablations of gemm_xglds.hip (VMV_XGLDS_ABLATE builds in ab_libs/xg<k>): 1 no MFMAs, 2 no LDS-DMA after the prologue,
# 3 no fragment reads, 4 no block barriers — conv / temporal conv shapes of the two large levels, tile 20 forced
#   for k in 1 2 3 4; do make -C videomv_amd/csrc EXTRA=-DVMV_XGLDS_ABLATE=$k LIBDIR=../../ab_libs/xg$k BUILD=build_xg$k ../../ab_libs/xg$k/libvmv_hip_f16.so; done
echo "== product"; VMV_BENCH_SHAPES="conv L0,tcnv L0,conv L1" python tools/gemm_bench.py 20
for k in 1 2 3 4; do echo "== ablate $k"; VMV_LIB_DIR=$PWD/ab_libs/xg$k VMV_BENCH_SHAPES="conv L0,tcnv L0,conv L1" python tools/gemm_bench.py 20; done

# attention ablations (attention.hip VMV_ATTN_ABLATE, libraries built into ab_libs/attn<k>): what bounds attn_kernel<4,4>?
#   for k in 1 2 3 4 5; do make -C videomv_amd/csrc EXTRA=-DVMV_ATTN_ABLATE=$k LIBDIR=../../ab_libs/attn$k BUILD=build_attn$k ../../ab_libs/attn$k/libvmv_hip_f16.so; done
echo "== product"; python tools/attn_bench.py
for k in 1 2 3 4 5; do echo "== ablate $k"; VMV_LIB_DIR=$PWD/ab_libs/attn$k python tools/attn_bench.py | head -2; done

export VMV_BENCH_SHAPES="L0 N960,L0 N2560"
for ab in 0 7 1; do echo "== ablate $ab"; VMV_GEMM_ABLATE=$ab python tools/gemm_bench.py 18 19 2>&1 | grep -v amdgpu.ids; done

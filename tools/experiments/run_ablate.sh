# ablation matrix of the persistent short-K GEMM on the L0 / L1 transformer shapes (one box, same session)
export VMV_BENCH_SHAPES="L0 N320 K320,L0 N960,L0 N2560,down L0,L1 N1920,L1 N5120"
for ab in 0 1 2 7 3 4; do
  echo "=== VMV_GEMM_ABLATE=$ab  (0 full, 1 no MFMA/frag reads, 2 no DMA, 7 no stores, 3 geglu without erf, 4 stamps)"
  VMV_GEMM_ABLATE=$ab python tools/gemm_bench.py 0 9 10 2>&1 | grep -v amdgpu.ids
done

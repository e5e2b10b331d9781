python -m pytest tests/test_kernels_gpu.py -q -k "layernorm_folded or astat or groupnorm" 2>&1 | tail -15 > gpurun_out/r2c_pytest.log
cat gpurun_out/r2c_pytest.log
export VMV_BENCH_SHAPES="L0 N960,L0 N2560"
python tools/gemm_bench.py 0 9 10 18 19 2>&1 | grep -v amdgpu.ids > gpurun_out/r2c_gemm.log
cat gpurun_out/r2c_gemm.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sample --no-lgm --no-op-profile"
VMV_GEMM_ASTAT=0 VMV_GN_FUSED=0 $B > gpurun_out/r2c_base.json 2> gpurun_out/r2c.err
VMV_GEMM_ASTAT=0 VMV_GN_FUSED=1 $B > gpurun_out/r2c_gn.json 2>> gpurun_out/r2c.err
VMV_GEMM_ASTAT=1 VMV_GN_FUSED=1 $B > gpurun_out/r2c_gn_astat.json 2>> gpurun_out/r2c.err
VMV_GEMM_ASTAT=1 VMV_GN_FUSED=0 $B > gpurun_out/r2c_astat.json 2>> gpurun_out/r2c.err
python -c "
import json
for n in ('base','gn','gn_astat','astat'):
    d=json.load(open(f'gpurun_out/r2c_{n}.json')); print(n, d['value'], d['ms_per_step'], d['finite'])
"

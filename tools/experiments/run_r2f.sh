python -m pytest tests/test_kernels_gpu.py tests/test_unet_gpu.py tests/test_gs_gpu.py -q -k "attention or lgm or vae or splitk or split" 2>&1 | tail -8
VMV_OUT=gpurun_out python tools/experiments/lgm_step_bench.py 2>&1 | grep -v amdgpu.ids | tail -6

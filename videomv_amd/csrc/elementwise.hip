// elementwise.hip — sampler glue kernels (HBM/launch-latency bound, tiny): layout moves between the
// reference's NCFHW fp32 latents and the channels-last bf16 rows, the fused CFG + DDIM update, and the
// time/camera embedding combine.  gfx950, fp32 math.
#include "common.h"

namespace {

__global__ void latent_to_rows_kernel(const float* __restrict__ x, uint16_t* __restrict__ rows, int nb, int C, int F,
                                      long HW, int Cpad, int nrep) {
    // one thread per (b, f, pixel): gathers C strided floats, writes Cpad bf16 contiguous
    const long per = (long)nb * F * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (long)gridDim.x * blockDim.x) {
        const long pix = i % HW;
        const long bf = i / HW;
        const int f = (int)(bf % F);
        const int b = (int)(bf / F);
        for (int rep = 0; rep < nrep; ++rep) {
            uint16_t* dst = rows + ((long)rep * per + i) * Cpad;
            for (int c = 0; c < Cpad; ++c) {
                const float v = (c < C) ? x[(((long)b * C + c) * F + f) * HW + pix] : 0.f;
                dst[c] = (uint16_t)f32_to_bf16_bits(v);
            }
        }
    }
}

__global__ void rows_to_nchw_kernel(const void* __restrict__ rows, int rows_fp32, int ld, float* __restrict__ out, long n,
                                    int C, long HW) {
    const long total = n * C * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i % HW;
        const long t = i / HW;
        const int c = (int)(t % C);
        const long img = t / C;
        const long src = (img * HW + pix) * ld + c;
        out[i] = rows_fp32 ? reinterpret_cast<const float*>(rows)[src]
                           : bf16_to_f32(reinterpret_cast<const uint16_t*>(rows)[src]);
    }
}

// xt layout [C][F][HW] (batch 1, the reference's noise shape [1,4,F,h,w]); eps rows [2][F*HW][ld].
__global__ void cfg_ddim_kernel(const VmvDdimParams p) {
    const long FHW = (long)p.F * p.HW;
    const long total = (long)p.C * FHW;
    const float sqrt_aprev = sqrtf(p.a_prev);
    const float sqrt_1m = sqrtf(1.0f - p.a_prev);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i / FHW);
        const long r = i - (long)c * FHW;
        const float y = p.eps_rows[r * p.ld + c];
        const float u = p.eps_rows[(FHW + r) * p.ld + c];
        const float out = u + p.guide_scale * (y - u);           // diffusion_ddim.py:157-160
        const float xt = p.xt[i];
        float x0;
        if (p.v_pred) x0 = p.c_sqrt_ac * xt - p.c_sqrt_1mac * out;   // :196-199
        else x0 = p.c_recip * xt - p.c_recipm1 * out;                // :192-195
        const float eps = (p.c_recip * xt - x0) / p.c_recipm1;       // :233-234
        p.xt[i] = sqrt_aprev * x0 + sqrt_1m * eps;                   // :240-243 (eta = 0 -> sigma = 0)
        if (p.x0_out) p.x0_out[i] = x0;
    }
}

__global__ void emb_combine_kernel(const float* __restrict__ temb, const float* __restrict__ cam, uint16_t* __restrict__ out,
                                   int rows, int C, int rows_per_t, int cam_rows) {
    const long total = (long)rows * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / C), c = (int)(i % C);
        float v = temb[(long)(r / rows_per_t) * C + c];
        if (cam) v += cam[(long)(r % cam_rows) * C + c];
        out[i] = (uint16_t)f32_to_bf16_bits(silu_f(v));
    }
}

__global__ void sinusoidal_kernel(const float* __restrict__ t, uint16_t* __restrict__ out, int n, int dim) {
    const int half = dim >> 1;
    const int total = n * dim;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int r = i / dim, c = i % dim;
        float v = 0.f;
        if (c < 2 * half) {
            const int k = c < half ? c : c - half;
            const float freq = powf(10000.0f, -(float)k / (float)half);
            const float a = t[r] * freq;
            v = c < half ? cosf(a) : sinf(a);   // cos first (util.py:186)
        }
        out[i] = (uint16_t)f32_to_bf16_bits(v);
    }
}

inline int grid_for(long n, int block = 256, int cap = 8192) {
    long g = (n + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int vmv_latent_to_rows(const float* x, void* rows, int nb, int C, int F, int H, int W, int Cpad, int nrep,
                                  void* stream) {
    if (!x || !rows) return VMV_ENULL;
    if (nb <= 0 || C <= 0 || F <= 0 || H <= 0 || W <= 0 || Cpad < C || nrep <= 0) return VMV_EINVAL;
    const long HW = (long)H * W;
    hipLaunchKernelGGL(latent_to_rows_kernel, dim3(grid_for((long)nb * F * HW)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), x, reinterpret_cast<uint16_t*>(rows), nb, C, F, HW, Cpad, nrep);
    return vmv_launch_status();
}

extern "C" int vmv_rows_to_nchw(const void* rows, int rows_fp32, int ld, float* out, int n, int C, int HW, void* stream) {
    if (!rows || !out) return VMV_ENULL;
    if (n <= 0 || C <= 0 || HW <= 0 || ld < C) return VMV_EINVAL;
    hipLaunchKernelGGL(rows_to_nchw_kernel, dim3(grid_for((long)n * C * HW)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), rows, rows_fp32, ld, out, (long)n, C, (long)HW);
    return vmv_launch_status();
}

extern "C" int vmv_cfg_ddim_step(const VmvDdimParams* pp, void* stream) {
    if (!pp) return VMV_ENULL;
    const VmvDdimParams& p = *pp;
    if (!p.eps_rows || !p.xt) return VMV_ENULL;
    if (p.C <= 0 || p.F <= 0 || p.HW <= 0 || p.ld < p.C) return VMV_EINVAL;
    hipLaunchKernelGGL(cfg_ddim_kernel, dim3(grid_for((long)p.C * p.F * p.HW)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), p);
    return vmv_launch_status();
}

extern "C" int vmv_emb_combine_silu(const float* temb, const float* cam, void* out, int rows, int C, int rows_per_t,
                                    int cam_rows, void* stream) {
    if (!temb || !out) return VMV_ENULL;
    if (rows <= 0 || C <= 0 || rows_per_t <= 0 || (cam && cam_rows <= 0)) return VMV_EINVAL;
    hipLaunchKernelGGL(emb_combine_kernel, dim3(grid_for((long)rows * C)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), temb, cam, reinterpret_cast<uint16_t*>(out), rows, C,
                       rows_per_t, cam_rows > 0 ? cam_rows : 1);
    return vmv_launch_status();
}

extern "C" int vmv_sinusoidal(const float* t, void* out_bf16, int n, int dim, void* stream) {
    if (!t || !out_bf16) return VMV_ENULL;
    if (n <= 0 || dim <= 0) return VMV_EINVAL;
    hipLaunchKernelGGL(sinusoidal_kernel, dim3(grid_for((long)n * dim)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), t, reinterpret_cast<uint16_t*>(out_bf16), n, dim);
    return vmv_launch_status();
}

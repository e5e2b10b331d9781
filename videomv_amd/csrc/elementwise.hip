// elementwise.hip — sampler glue kernels (HBM/launch-latency bound, tiny): layout moves between the
// reference's NCFHW fp32 latents and the channels-last bf16 rows, the fused CFG + DDIM update, and the
// time/camera embedding combine.  gfx950, fp32 math.
#include "common.h"

namespace {

__global__ void latent_to_rows_kernel(const float* __restrict__ x, uint16_t* __restrict__ rows, int nb, int C, int F,
                                      long HW, int Cpad, int nrep) {
    VMV_KERNEL_ENTER();
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
                dst[c] = (uint16_t)f32_to_elem_bits(v);
            }
        }
    }
}

__global__ void latent_to_rows_keep_kernel(const float* __restrict__ x, uint16_t* __restrict__ rows, int nb, int C, int F,
                                           long HW, int ld, int nrep) {
    VMV_KERNEL_ENTER();
    const long per = (long)nb * F * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (long)gridDim.x * blockDim.x) {
        const long pix = i % HW;
        const long bf = i / HW;
        const int f = (int)(bf % F);
        const int b = (int)(bf / F);
        for (int rep = 0; rep < nrep; ++rep) {
            uint16_t* dst = rows + ((long)rep * per + i) * ld;
            for (int c = 0; c < C; ++c) dst[c] = (uint16_t)f32_to_elem_bits(x[(((long)b * C + c) * F + f) * HW + pix]);
        }
    }
}

__global__ void lgm_x0_views_kernel(const float* __restrict__ eps, int ld, int branch, const float* __restrict__ xt, int C,
                                    int F, long HW, int i0, int i1, int i2, int i3, float cr, float crm1, float inv_scale,
                                    float* __restrict__ out) {
    VMV_KERNEL_ENTER();
    const long total = 4L * C * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i % HW;
        const int c = (int)((i / HW) % C);
        const int v = (int)(i / (HW * C));
        const int f = v == 0 ? i0 : (v == 1 ? i1 : (v == 2 ? i2 : i3));
        const float e = eps[(((long)branch * F + f) * HW + pix) * ld + c];
        out[i] = inv_scale * (cr * xt[((long)c * F + f) * HW + pix] - crm1 * e);
    }
}

__global__ void lgm_pack_input_kernel(const float* __restrict__ dec, const float* __restrict__ rays, float* __restrict__ out,
                                      int nviews, long HW) {
    VMV_KERNEL_ENTER();
    const long total = (long)nviews * 9 * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i % HW;
        const int c = (int)((i / HW) % 9);
        const long v = i / (HW * 9);
        if (c < 3) {
            const float mean = c == 0 ? 0.485f : (c == 1 ? 0.456f : 0.406f);
            const float stdv = c == 0 ? 0.229f : (c == 1 ? 0.224f : 0.225f);
            const float x = fminf(1.f, fmaxf(0.f, dec[(v * 3 + c) * HW + pix] * 0.5f + 0.5f));
            out[i] = (x - mean) / stdv;
        } else {
            out[i] = rays[(v * 6 + (c - 3)) * HW + pix];
        }
    }
}

// nearest resampling as F.interpolate(mode='nearest') does it: source index = floor(dst * in / out)
__global__ void lgm_render_to_vae_kernel(const float* __restrict__ img, float* __restrict__ out, int nviews, int S_in, int S) {
    VMV_KERNEL_ENTER();
    const long total = (long)nviews * 3 * S * S;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % S), y = (int)((i / S) % S);
        const long vc = i / ((long)S * S);
        const int sx = (int)(((long)x * S_in) / S), sy = (int)(((long)y * S_in) / S);
        out[i] = (img[(vc * S_in + sy) * S_in + sx] - 0.5f) / 0.5f;
    }
}

__global__ void ddim_x0_step_kernel(const float* __restrict__ xc, const float* __restrict__ xu, float* __restrict__ xt, long n,
                                    float guide, float cr, float crm1, float a_prev, float clamp, float sigma,
                                    const float* __restrict__ noise) {
    VMV_KERNEL_ENTER();
    const float sa = sqrtf(a_prev), sb = sqrtf(1.0f - a_prev - sigma * sigma);   // diffusion_ddim.py:241
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float x0 = xu[i] + guide * (xc[i] - xu[i]);                              // :157-160 on latent_z, x0 = out (:179-182)
        if (clamp > 0.f) x0 = fminf(fmaxf(x0, -clamp), clamp);                   // :204-205 (applied on refined steps too)
        const float eps = (cr * xt[i] - x0) / crm1;
        float nx = sa * x0 + sb * eps;
        if (sigma > 0.f) nx += sigma * noise[i];                                 // :240-243
        xt[i] = nx;
    }
}

// LGM Gaussian head.  Pass 1: per-block sums of squares of the 4 rotation components (fixed order: strided rows per
// thread, then a tree over the block) -> workspace[block][4].  Pass 2: every block folds the <= 256 partials in the same
// order, then applies the activations to its rows.
__global__ __launch_bounds__(256) void gauss_rot_sumsq_kernel(const float* __restrict__ raw, int ld, int n, float* __restrict__ ws) {
    VMV_KERNEL_ENTER();
    __shared__ float sh[4][256];
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float* r = raw + i * ld + 7;
#pragma unroll
        for (int c = 0; c < 4; ++c) s[c] += r[c] * r[c];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) sh[c][threadIdx.x] = s[c];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
#pragma unroll
            for (int c = 0; c < 4; ++c) sh[c][threadIdx.x] += sh[c][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x < 4) ws[blockIdx.x * 4 + threadIdx.x] = sh[threadIdx.x][0];
}

__global__ __launch_bounds__(256) void gauss_activation_kernel(const float* __restrict__ raw, int ld, float* __restrict__ out,
                                                               int n, const float* __restrict__ ws, int nparts) {
    VMV_KERNEL_ENTER();
    __shared__ float inv[4];
    if (threadIdx.x < 4) {
        float t = 0.f;
        for (int b = 0; b < nparts; ++b) t += ws[b * 4 + threadIdx.x];
        inv[threadIdx.x] = 1.0f / fmaxf(sqrtf(t), 1e-12f);
    }
    __syncthreads();
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = raw + i * ld;
    float* o = out + i * 14;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = fminf(fmaxf(r[c], -1.f), 1.f);
    o[3] = 1.0f / (1.0f + expf(-r[3]));
#pragma unroll
    for (int c = 4; c < 7; ++c) o[c] = 0.1f * (r[c] > 20.f ? r[c] : log1pf(expf(r[c])));      // F.softplus (threshold 20)
#pragma unroll
    for (int c = 0; c < 4; ++c) o[7 + c] = r[7 + c] * inv[c];
#pragma unroll
    for (int c = 11; c < 14; ++c) o[c] = 0.5f * tanhf(r[c]) + 0.5f;
}

// dst (contiguous [n0][n1][n2][inner16] of 16-byte vectors) <- src with per-axis strides: both sides move whole
// channel rows, so every lane does 16-byte loads and stores and consecutive lanes touch consecutive vectors.
__global__ __launch_bounds__(256) void permute_copy_kernel(const VmvCopyParams p, const long total) {
    VMV_KERNEL_ENTER();
    const u32x4_t* __restrict__ src = reinterpret_cast<const u32x4_t*>(p.src);
    u32x4_t* __restrict__ dst = reinterpret_cast<u32x4_t*>(p.dst);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long v = i % p.inner16;
        long blk = i / p.inner16;
        const long i2 = blk % p.n2; blk /= p.n2;
        const long i1 = blk % p.n1;
        const long i0 = blk / p.n1;
        dst[i] = src[i0 * p.ss0 + i1 * p.ss1 + i2 * p.ss2 + v];
    }
}

// One wave per pixel, lane = frame (F <= 64).  Everything for a pixel lives in registers; k/v of other frames are
// fetched with wave shuffles.  Tiny (2560 pixels x ~10 kFLOP): latency only, runs once per sample.
__global__ __launch_bounds__(256) void i2v_temporal_adapter_kernel(const uint16_t* __restrict__ in, int ld_in,
                                                                   uint16_t* __restrict__ out, int ld_out,
                                                                   const float* __restrict__ w, int F, int HW, int nrep,
                                                                   float scale) {
    VMV_KERNEL_ENTER();
    const int lane = threadIdx.x & 63;
    const int pix = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= HW) return;
    const float* ln_g = w; const float* ln_b = w + 4; const float* Wqkv = w + 8; const float* Wo = w + 104;
    const float* bo = w + 136; const float* W1 = w + 140; const float* b1 = w + 204; const float* W2 = w + 220;
    const float* b2 = w + 284;
    const bool act = lane < F;
    float x[4] = {0.f, 0.f, 0.f, 0.f};
    if (act) {
        const uint16_t* src = in + ((long)lane * HW + pix) * ld_in;
#pragma unroll
        for (int c = 0; c < 4; ++c) x[c] = elem_to_f32(src[c]);
    }
    // LayerNorm over the 4 channels (eps 1e-5)
    const float mean = 0.25f * (x[0] + x[1] + x[2] + x[3]);
    float var = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) var += (x[c] - mean) * (x[c] - mean);
    const float rstd = rsqrtf(0.25f * var + 1e-5f);
    float hn[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) hn[c] = (x[c] - mean) * rstd * ln_g[c] + ln_b[c];
    float qkv[24];
#pragma unroll
    for (int o = 0; o < 24; ++o) {
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) a += Wqkv[o * 4 + c] * hn[c];
        qkv[o] = a;
    }
    // attention: heads 2, dim_head 4, scale 4^-0.5; q = qkv[0:8], k = qkv[8:16], v = qkv[16:24], head h -> [4h, 4h+4)
    float o8[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float m = -3.0e38f, l = 0.f, acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < F; ++j) {
            float s = 0.f, vj[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                s += qkv[4 * h + d] * __shfl(qkv[8 + 4 * h + d], j, 64);
                vj[d] = __shfl(qkv[16 + 4 * h + d], j, 64);
            }
            s *= 0.5f;
            const float mn = fmaxf(m, s);
            const float a = __expf(m - mn), e = __expf(s - mn);
            l = l * a + e;
#pragma unroll
            for (int d = 0; d < 4; ++d) acc[d] = acc[d] * a + e * vj[d];
            m = mn;
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) o8[4 * h + d] = acc[d] / l;
    }
    float y[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float a = bo[c];
#pragma unroll
        for (int k = 0; k < 8; ++k) a += Wo[c * 8 + k] * o8[k];
        y[c] = a + x[c];
    }
    float hid[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        float a = b1[k];
#pragma unroll
        for (int c = 0; c < 4; ++c) a += W1[k * 4 + c] * y[c];
        hid[k] = gelu_erf_f(a);
    }
    if (act) {
        uint16_t r[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a = b2[c];
#pragma unroll
            for (int k = 0; k < 16; ++k) a += W2[c * 16 + k] * hid[k];
            r[c] = (uint16_t)f32_to_elem_bits(scale * (a + y[c]));
        }
        for (int rep = 0; rep < nrep; ++rep) {
            uint16_t* dst = out + (((long)rep * F + lane) * HW + pix) * ld_out;
#pragma unroll
            for (int c = 0; c < 4; ++c) dst[c] = r[c];
        }
    }
}

__global__ void adaptive_avgpool_rows_kernel(const uint16_t* __restrict__ in, int ld, uint16_t* __restrict__ out, int ldo,
                                             int n, int C, int IH, int IW, int OH, int OW) {
    VMV_KERNEL_ENTER();
    const long total = (long)n * OH * OW * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long t = i / C;
        const int ox = (int)(t % OW); t /= OW;
        const int oy = (int)(t % OH);
        const int img = (int)(t / OH);
        // PyTorch bins: [floor(o*I/O), ceil((o+1)*I/O))
        const int y0 = (oy * IH) / OH, y1 = ((oy + 1) * IH + OH - 1) / OH;
        const int x0 = (ox * IW) / OW, x1 = ((ox + 1) * IW + OW - 1) / OW;
        float s = 0.f;
        for (int yy = y0; yy < y1; ++yy)
            for (int xx = x0; xx < x1; ++xx) s += elem_to_f32(in[(((long)img * IH + yy) * IW + xx) * ld + c]);
        out[(((long)img * OH + oy) * OW + ox) * ldo + c] = (uint16_t)f32_to_elem_bits(s / (float)((y1 - y0) * (x1 - x0)));
    }
}

__global__ void rows_to_nchw_kernel(const void* __restrict__ rows, int rows_fp32, int ld, float* __restrict__ out, long n,
                                    int C, long HW) {
    VMV_KERNEL_ENTER();
    const long total = n * C * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i % HW;
        const long t = i / HW;
        const int c = (int)(t % C);
        const long img = t / C;
        const long src = (img * HW + pix) * ld + c;
        out[i] = rows_fp32 ? reinterpret_cast<const float*>(rows)[src]
                           : elem_to_f32(reinterpret_cast<const uint16_t*>(rows)[src]);
    }
}

// xt layout [C][F][HW] (batch 1, the reference's noise shape [1,4,F,h,w]); eps rows [2][F*HW][ld].
__global__ void cfg_ddim_kernel(const VmvDdimParams p) {
    VMV_KERNEL_ENTER();
    const long FHW = (long)p.F * p.HW;
    const long total = (long)p.C * FHW;
    const float sqrt_aprev = sqrtf(p.a_prev);
    const float sqrt_1m = sqrtf(1.0f - p.a_prev - p.sigma * p.sigma);     // :241 (sigma = 0 unless eta > 0)
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
        if (p.clamp > 0.f) x0 = fminf(fmaxf(x0, -p.clamp), p.clamp); // :204-205
        const float eps = (p.c_recip * xt - x0) / p.c_recipm1;       // :233-234
        float nx = sqrt_aprev * x0 + sqrt_1m * eps;                  // :240-243
        if (p.sigma > 0.f) nx += p.sigma * p.noise[i];               // (mask = t != 0 is always 1: the DDIM steps start at 1)
        p.xt[i] = nx;
        if (p.x0_out) p.x0_out[i] = x0;
    }
}

__global__ void posterior_sample_kernel(const float* __restrict__ mom, int ld, const float* __restrict__ noise,
                                        float* __restrict__ z, long n, int zc, long HW, float scale) {
    VMV_KERNEL_ENTER();
    const long total = n * zc * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i % HW;
        const long t = i / HW;
        const int c = (int)(t % zc);
        const long img = t / zc;
        const float* m = mom + (img * HW + pix) * ld;
        const float lv = fminf(fmaxf(m[zc + c], -30.0f), 20.0f);
        z[i] = scale * (m[c] + __expf(0.5f * lv) * noise[i]);
    }
}

__global__ void emb_combine_kernel(const float* __restrict__ temb, const float* __restrict__ cam, uint16_t* __restrict__ out,
                                   int rows, int C, int rows_per_t, int cam_rows) {
    VMV_KERNEL_ENTER();
    const long total = (long)rows * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / C), c = (int)(i % C);
        float v = temb[(long)(r / rows_per_t) * C + c];
        if (cam) v += cam[(long)(r % cam_rows) * C + c];
        out[i] = (uint16_t)f32_to_elem_bits(silu_f(v));
    }
}

__global__ void sinusoidal_kernel(const float* __restrict__ t, uint16_t* __restrict__ out, int n, int dim) {
    VMV_KERNEL_ENTER();
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
        out[i] = (uint16_t)f32_to_elem_bits(v);
    }
}

inline int grid_for(long n, int block = 256, int cap = 8192) {
    long g = (n + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int vmv_lgm_x0_views(const float* eps_rows, int ld, int branch, const float* xt, int C, int F, int HW,
                                const int32_t* idx4, float c_recip, float c_recipm1, float inv_scale, float* out, void* stream) {
    if (!eps_rows || !xt || !idx4 || !out) return VMV_ENULL;
    if (ld < C || C <= 0 || F <= 0 || HW <= 0 || branch < 0) return VMV_EINVAL;
    for (int v = 0; v < 4; ++v) if (idx4[v] < 0 || idx4[v] >= F) return VMV_ERANGE;
    hipLaunchKernelGGL(lgm_x0_views_kernel, dim3(grid_for(4L * C * HW)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       eps_rows, ld, branch, xt, C, F, (long)HW, idx4[0], idx4[1], idx4[2], idx4[3], c_recip, c_recipm1,
                       inv_scale, out);
    return vmv_launch_status();
}

extern "C" int vmv_lgm_pack_input(const float* decoded, const float* rays, float* out, int nviews, int HW, void* stream) {
    if (!decoded || !rays || !out) return VMV_ENULL;
    if (nviews <= 0 || HW <= 0) return VMV_EINVAL;
    hipLaunchKernelGGL(lgm_pack_input_kernel, dim3(grid_for((long)nviews * 9 * HW)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), decoded, rays, out, nviews, (long)HW);
    return vmv_launch_status();
}

extern "C" int vmv_lgm_render_to_vae(const float* images, float* out, int nviews, int S_in, int S, void* stream) {
    if (!images || !out) return VMV_ENULL;
    if (nviews <= 0 || S <= 0 || S_in <= 0) return VMV_EINVAL;
    hipLaunchKernelGGL(lgm_render_to_vae_kernel, dim3(grid_for((long)nviews * 3 * S * S)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), images, out, nviews, S_in, S);
    return vmv_launch_status();
}

extern "C" int vmv_ddim_x0_step(const float* x0_cond, const float* x0_uncond, float* xt, long n, float guide, float c_recip,
                                float c_recipm1, float a_prev, float clamp, float sigma, const float* noise, void* stream) {
    if (!x0_cond || !x0_uncond || !xt) return VMV_ENULL;
    if (sigma > 0.0f && !noise) return VMV_ENULL;
    if (n <= 0 || c_recipm1 == 0.0f || clamp < 0.0f || sigma < 0.0f || 1.0f - a_prev - sigma * sigma < 0.0f) return VMV_EINVAL;
    hipLaunchKernelGGL(ddim_x0_step_kernel, dim3(grid_for(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x0_cond,
                       x0_uncond, xt, n, guide, c_recip, c_recipm1, a_prev, clamp, sigma, noise);
    return vmv_launch_status();
}

extern "C" int vmv_gaussian_activation(const float* raw, int ld, float* out, int n, float* workspace, void* stream) {
    if (!raw || !out || !workspace) return VMV_ENULL;
    if (n <= 0 || ld < 14) return VMV_EINVAL;
    int blocks = (n + 255) / 256;
    if (blocks > 256) blocks = 256;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(gauss_rot_sumsq_kernel, dim3(blocks), dim3(256), 0, st, raw, ld, n, workspace);
    hipLaunchKernelGGL(gauss_activation_kernel, dim3((n + 255) / 256), dim3(256), 0, st, raw, ld, out, n, workspace, blocks);
    return vmv_launch_status();
}

extern "C" int vmv_permute_copy(const VmvCopyParams* pp, void* stream) {
    if (!pp) return VMV_ENULL;
    const VmvCopyParams& p = *pp;
    if (!p.src || !p.dst) return VMV_ENULL;
    if (p.n0 < 1 || p.n1 < 1 || p.n2 < 1 || p.inner16 < 1 || p.ss0 < 0 || p.ss1 < 0 || p.ss2 < 0) return VMV_EINVAL;
    if (!vmv_aligned16(p.src) || !vmv_aligned16(p.dst)) return VMV_EALIGN;
    const long total = (long)p.n0 * p.n1 * p.n2 * p.inner16;
    long blocks = (total + 1023) / 1024;      // 4 vectors per thread
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(permute_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, total);
    return vmv_launch_status();
}

extern "C" int vmv_latent_to_rows(const float* x, void* rows, int nb, int C, int F, int H, int W, int Cpad, int nrep,
                                  void* stream) {
    if (!x || !rows) return VMV_ENULL;
    if (nb <= 0 || C <= 0 || F <= 0 || H <= 0 || W <= 0 || Cpad < C || nrep <= 0) return VMV_EINVAL;
    const long HW = (long)H * W;
    hipLaunchKernelGGL(latent_to_rows_kernel, dim3(grid_for((long)nb * F * HW)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), x, reinterpret_cast<uint16_t*>(rows), nb, C, F, HW, Cpad, nrep);
    return vmv_launch_status();
}

extern "C" int vmv_latent_to_rows_keep(const float* x, void* rows, int nb, int C, int F, int H, int W, int ld, int nrep,
                                       void* stream) {
    if (!x || !rows) return VMV_ENULL;
    if (nb <= 0 || C <= 0 || F <= 0 || H <= 0 || W <= 0 || ld < C || nrep <= 0) return VMV_EINVAL;
    const long HW = (long)H * W;
    hipLaunchKernelGGL(latent_to_rows_keep_kernel, dim3(grid_for((long)nb * F * HW)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), x, reinterpret_cast<uint16_t*>(rows), nb, C, F, HW, ld, nrep);
    return vmv_launch_status();
}

extern "C" int vmv_i2v_temporal_adapter(const void* in, int ld_in, void* out, int ld_out, const float* w, int F, int HW,
                                        int nrep, float scale, void* stream) {
    if (!in || !out || !w) return VMV_ENULL;
    if (F <= 0 || F > 64 || HW <= 0 || ld_in < 4 || ld_out < 4 || nrep <= 0) return VMV_EINVAL;
    hipLaunchKernelGGL(i2v_temporal_adapter_kernel, dim3((HW + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const uint16_t*>(in), ld_in, reinterpret_cast<uint16_t*>(out), ld_out, w, F, HW, nrep,
                       scale);
    return vmv_launch_status();
}

extern "C" int vmv_adaptive_avgpool_rows(const void* in, int ld, void* out, int ldo, int n, int C, int IH, int IW, int OH,
                                         int OW, void* stream) {
    if (!in || !out) return VMV_ENULL;
    if (n <= 0 || C <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0 || ld < C || ldo < C) return VMV_EINVAL;
    hipLaunchKernelGGL(adaptive_avgpool_rows_kernel, dim3(grid_for((long)n * OH * OW * C)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const uint16_t*>(in), ld,
                       reinterpret_cast<uint16_t*>(out), ldo, n, C, IH, IW, OH, OW);
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
    if (p.sigma < 0.f || p.clamp < 0.f || p.sigma * p.sigma > 1.0f - p.a_prev) return VMV_EINVAL;
    if (p.sigma > 0.f && !p.noise) return VMV_ENULL;
    hipLaunchKernelGGL(cfg_ddim_kernel, dim3(grid_for((long)p.C * p.F * p.HW)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), p);
    return vmv_launch_status();
}

extern "C" int vmv_posterior_sample(const float* moments_rows, int ld, const float* noise, float* z, int n, int zc, int HW,
                                    float scale, void* stream) {
    if (!moments_rows || !noise || !z) return VMV_ENULL;
    if (n <= 0 || zc <= 0 || HW <= 0 || ld < 2 * zc) return VMV_EINVAL;
    hipLaunchKernelGGL(posterior_sample_kernel, dim3(grid_for((long)n * zc * HW)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), moments_rows, ld, noise, z, (long)n, zc, (long)HW, scale);
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

extern "C" int vmv_sinusoidal(const float* t, void* out_elem, int n, int dim, void* stream) {
    if (!t || !out_elem) return VMV_ENULL;
    if (n <= 0 || dim <= 0) return VMV_EINVAL;
    hipLaunchKernelGGL(sinusoidal_kernel, dim3(grid_for((long)n * dim)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), t, reinterpret_cast<uint16_t*>(out_elem), n, dim);
    return vmv_launch_status();
}

// raster.hip — forward Gaussian-splatting rasteriser for the LGM refinement branch (gfx950).
// Replaces the reference's call into the third-party `diff_gaussian_rasterization` extension (core/gs.py:57-83:
// sh_degree 0, colours precomputed, scale_modifier 1, square image).  That extension is absent from the reference tree
// and unpinned (install.sh:3), so the algorithm follows the published 3-D Gaussian Splatting forward pass as restated in
// oracle/gs_ref.py (parity unpinned; validated on analytic cases and against that restatement).
//
// Pipeline per view (all on the caller's stream):
//   gs_preprocess_kernel   one thread per Gaussian: view/clip transform, near cull, 3-D covariance from (scale, raw
//                          quaternion), EWA 2-D covariance (+0.3 I), conic, 3-sigma radius, 16x16-tile rectangle,
//                          tiles touched
//   rocprim::inclusive_scan                      offsets of every Gaussian's (tile, depth) instances
//   (host reads the total, sizes the binning buffers)
//   gs_duplicate_kernel    key = tile << 32 | depth bits, value = Gaussian index
//   rocprim::radix_sort_pairs                    (the only library primitive: a stable 64-bit key sort)
//   gs_ranges_kernel       first / last instance of every tile
//   gs_render_kernel       one 256-thread block per 16x16 tile: the tile's instances are staged through LDS in batches
//                          of 256 (xy, conic+opacity, colour), every thread blends its pixel front to back with the
//                          1/255 alpha cut, the 0.99 clamp and the T < 1e-4 stop; block-wide early exit
// Memory: everything per Gaussian is 48 B, per instance 12 B; the render pass reads each instance once per tile.
#include "common.h"
#include <cstring>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_radix_sort.hpp>

namespace {

constexpr int GS_TILE = 16;

// One Gaussian against one view: everything gs_preprocess writes for it.  `V` / `M` = the view's cam_view / cam_view_proj (row-major,
// row-vector convention), outputs at index `o` of the per-(view, Gaussian) arrays.  Shared by the per-view and the batched kernels, so
// the two paths produce the same bits.
struct GsOut { float* depth; float* xy; float* conic_opacity; int32_t* rect; uint32_t* tiles_touched; };
VMV_DEV void gs_project(const float* __restrict__ g, const float* __restrict__ V, const float* __restrict__ M, const int size,
                        const float tan_half_fov, const GsOut& out, const long o) {
    out.tiles_touched[o] = 0;
    const float mx = g[0], my = g[1], mz = g[2];
    // row-vector convention: [m, 1] @ matrix
    const float vx = mx * V[0] + my * V[4] + mz * V[8] + V[12];
    const float vy = mx * V[1] + my * V[5] + mz * V[9] + V[13];
    const float vz = mx * V[2] + my * V[6] + mz * V[10] + V[14];
    if (!(vz > 0.2f)) return;
    const float hx = mx * M[0] + my * M[4] + mz * M[8] + M[12];
    const float hy = mx * M[1] + my * M[5] + mz * M[9] + M[13];
    const float hw = mx * M[3] + my * M[7] + mz * M[11] + M[15];
    const float iw = 1.0f / (hw + 1e-7f);
    const float nx = hx * iw, ny = hy * iw;
    // 3-D covariance  R diag(s^2) R^T, quaternion (r, x, y, z) used as given
    const float sx = g[4], sy = g[5], sz = g[6];
    const float qr = g[7], qx = g[8], qy = g[9], qz = g[10];
    const float R[9] = {1.f - 2.f * (qy * qy + qz * qz), 2.f * (qx * qy - qr * qz), 2.f * (qx * qz + qr * qy),
                        2.f * (qx * qy + qr * qz), 1.f - 2.f * (qx * qx + qz * qz), 2.f * (qy * qz - qr * qx),
                        2.f * (qx * qz - qr * qy), 2.f * (qy * qz + qr * qx), 1.f - 2.f * (qx * qx + qy * qy)};
    const float s2[3] = {sx * sx, sy * sy, sz * sz};
    float S3[9];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
            S3[a * 3 + b] = R[a * 3 + 0] * s2[0] * R[b * 3 + 0] + R[a * 3 + 1] * s2[1] * R[b * 3 + 1] + R[a * 3 + 2] * s2[2] * R[b * 3 + 2];
    // EWA projection: T = J W3, W3 = V[:3,:3]^T
    const float focal = (float)size / (2.0f * tan_half_fov);
    const float lim = 1.3f * tan_half_fov;
    const float tx = fminf(lim, fmaxf(-lim, vx / vz)) * vz;
    const float ty = fminf(lim, fmaxf(-lim, vy / vz)) * vz;
    const float j00 = focal / vz, j02 = -focal * tx / (vz * vz), j11 = focal / vz, j12 = -focal * ty / (vz * vz);
    float T0[3], T1[3];          // rows of T = J W3:  W3[r][c] = V[c*4 + r]
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        T0[c] = j00 * V[c * 4 + 0] + j02 * V[c * 4 + 2];
        T1[c] = j11 * V[c * 4 + 1] + j12 * V[c * 4 + 2];
    }
    float u0[3], u1[3];          // Sigma3 T^T
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        u0[a] = S3[a * 3 + 0] * T0[0] + S3[a * 3 + 1] * T0[1] + S3[a * 3 + 2] * T0[2];
        u1[a] = S3[a * 3 + 0] * T1[0] + S3[a * 3 + 1] * T1[1] + S3[a * 3 + 2] * T1[2];
    }
    const float ca = T0[0] * u0[0] + T0[1] * u0[1] + T0[2] * u0[2] + 0.3f;
    const float cb = T0[0] * u1[0] + T0[1] * u1[1] + T0[2] * u1[2];
    const float cc = T1[0] * u1[0] + T1[1] * u1[1] + T1[2] * u1[2] + 0.3f;
    const float det = ca * cc - cb * cb;
    if (det == 0.0f) return;
    const float idet = 1.0f / det;
    const float mid = 0.5f * (ca + cc);
    const float lam = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
    const float radius = ceilf(3.0f * sqrtf(lam));
    const float px = ((nx + 1.0f) * (float)size - 1.0f) * 0.5f;
    const float py = ((ny + 1.0f) * (float)size - 1.0f) * 0.5f;
    const int grid = (size + GS_TILE - 1) / GS_TILE;
    auto clampi = [&](float v) { const int t = (int)v; return t < 0 ? 0 : (t > grid ? grid : t); };
    const int x0 = clampi((px - radius) / GS_TILE), y0 = clampi((py - radius) / GS_TILE);
    const int x1 = clampi((px + radius + GS_TILE - 1) / GS_TILE), y1 = clampi((py + radius + GS_TILE - 1) / GS_TILE);
    const int touched = (x1 - x0) * (y1 - y0);
    if (touched <= 0) return;
    out.depth[o] = vz;
    out.xy[2 * o] = px; out.xy[2 * o + 1] = py;
    float* co = out.conic_opacity + 4L * o;
    co[0] = cc * idet; co[1] = -cb * idet; co[2] = ca * idet; co[3] = g[3];
    int* rc = out.rect + 4L * o;
    rc[0] = x0; rc[1] = y0; rc[2] = x1; rc[3] = y1;
    out.tiles_touched[o] = (uint32_t)touched;
}

__global__ __launch_bounds__(256) void gs_preprocess_kernel(const VmvGsParams p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.N) return;
    const GsOut out{p.depth, p.xy, p.conic_opacity, p.rect, p.tiles_touched};
    gs_project(p.gaussians + (long)i * 14, p.view, p.view_proj, p.size, p.tan_half_fov, out, i);
}

// ---- batched form (vmv.h VmvGsBatchParams): all B * V views of B samples in ONE pass — one preprocess launch, one scan, one
//      radix sort over every (view, tile, depth) instance, one ranges launch, one blend launch; "view" vv = b * V + v everywhere.
__global__ __launch_bounds__(256) void gs_preprocess_batch_kernel(const VmvGsBatchParams p) {
    __shared__ float s_m[32];
    const int vv = blockIdx.y;
    if (threadIdx.x < 16) s_m[threadIdx.x] = p.views[16 * vv + threadIdx.x];
    else if (threadIdx.x < 32) s_m[threadIdx.x] = p.view_projs[16 * vv + threadIdx.x - 16];
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.N) return;
    const GsOut out{p.depth, p.xy, p.conic_opacity, p.rect, p.tiles_touched};
    gs_project(p.gaussians + ((long)(vv / p.V) * p.N + i) * 14, s_m, s_m + 16, p.size, p.tan_half_fov, out, (long)vv * p.N + i);
}

__global__ __launch_bounds__(256) void gs_duplicate_batch_kernel(const VmvGsBatchParams p) {
    const int vv = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.N) return;
    const long o = (long)vv * p.N + i;
    if (p.tiles_touched[o] == 0) return;
    uint32_t off = o == 0 ? 0u : p.offsets[o - 1];
    const int* rc = p.rect + 4L * o;
    const int grid = (p.size + GS_TILE - 1) / GS_TILE;
    const uint32_t tile0 = (uint32_t)vv * (uint32_t)(grid * grid);
    const uint32_t dbits = __float_as_uint(p.depth[o]);           // depth > 0.2: the bit pattern orders like the float
    for (int y = rc[1]; y < rc[3]; ++y)
        for (int x = rc[0]; x < rc[2]; ++x) {
            if (off >= (uint32_t)p.num_rendered) return;
            p.keys[off] = ((uint64_t)(tile0 + (uint32_t)(y * grid + x)) << 32) | dbits;      // (view, tile) major, depth minor
            p.vals[off] = (uint32_t)i;                                                       // Gaussian index inside its sample
            ++off;
        }
}

__global__ __launch_bounds__(256) void gs_ranges_batch_kernel(const VmvGsBatchParams p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.num_rendered) return;
    const uint32_t t = (uint32_t)(p.keys_sorted[i] >> 32);
    if (i == 0) p.ranges[2 * t] = 0;
    else {
        const uint32_t tp = (uint32_t)(p.keys_sorted[i - 1] >> 32);
        if (tp != t) { p.ranges[2 * tp + 1] = (uint32_t)i; p.ranges[2 * t] = (uint32_t)i; }
    }
    if (i == p.num_rendered - 1) p.ranges[2 * t + 1] = (uint32_t)p.num_rendered;
}

__global__ __launch_bounds__(256) void gs_duplicate_kernel(const VmvGsParams p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.N || p.tiles_touched[i] == 0) return;
    uint32_t off = i == 0 ? 0u : p.offsets[i - 1];
    const int* rc = p.rect + 4L * i;
    const int grid = (p.size + GS_TILE - 1) / GS_TILE;
    const uint32_t dbits = __float_as_uint(p.depth[i]);           // depth > 0.2: the bit pattern orders like the float
    for (int y = rc[1]; y < rc[3]; ++y)
        for (int x = rc[0]; x < rc[2]; ++x) {
            if (off >= (uint32_t)p.num_rendered) return;         // (cannot happen when the host sized the buffers from offsets[N-1])
            p.keys[off] = ((uint64_t)(uint32_t)(y * grid + x) << 32) | dbits;
            p.vals[off] = (uint32_t)i;
            ++off;
        }
}

__global__ __launch_bounds__(256) void gs_ranges_kernel(const VmvGsParams p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.num_rendered) return;
    const uint32_t t = (uint32_t)(p.keys_sorted[i] >> 32);
    if (i == 0) p.ranges[2 * t] = 0;
    else {
        const uint32_t tp = (uint32_t)(p.keys_sorted[i - 1] >> 32);
        if (tp != t) { p.ranges[2 * tp + 1] = (uint32_t)i; p.ranges[2 * t] = (uint32_t)i; }
    }
    if (i == p.num_rendered - 1) p.ranges[2 * t + 1] = (uint32_t)p.num_rendered;
}

// One 16 x 16 tile of one view: blend the tile's sorted instances [lo, hi) front to back.  Arrays are the view's own (already offset).
VMV_DEV void gs_blend_tile(const int size, const float* __restrict__ bg, const uint32_t lo, const uint32_t hi,
                           const uint32_t* __restrict__ vals_sorted, const float* __restrict__ xy, const float* __restrict__ conic_opacity,
                           const float* __restrict__ gaussians, float* __restrict__ out_color, float* __restrict__ out_alpha) {
    __shared__ float s_xy[256][2];
    __shared__ float s_co[256][4];
    __shared__ float s_rgb[256][3];
    __shared__ int s_done;
    const int px = blockIdx.x * GS_TILE + (threadIdx.x & 15), py = blockIdx.y * GS_TILE + (threadIdx.x >> 4);
    const bool inside = px < size && py < size;
    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Wt = 0.f;
    bool done = !inside;
    for (uint32_t base = lo; base < hi; base += 256) {
        if (threadIdx.x == 0) s_done = 0;
        __syncthreads();
        const uint32_t j = base + threadIdx.x;
        if (j < hi) {
            const uint32_t gi = vals_sorted[j];
            s_xy[threadIdx.x][0] = xy[2 * gi]; s_xy[threadIdx.x][1] = xy[2 * gi + 1];
            const float* co = conic_opacity + 4L * gi;
            s_co[threadIdx.x][0] = co[0]; s_co[threadIdx.x][1] = co[1]; s_co[threadIdx.x][2] = co[2]; s_co[threadIdx.x][3] = co[3];
            const float* g = gaussians + 14L * gi + 11;
            s_rgb[threadIdx.x][0] = g[0]; s_rgb[threadIdx.x][1] = g[1]; s_rgb[threadIdx.x][2] = g[2];
        }
        __syncthreads();
        const int n = (int)min(256u, hi - base);
        for (int k = 0; k < n && !done; ++k) {
            const float dx = s_xy[k][0] - (float)px, dy = s_xy[k][1] - (float)py;
            const float power = -0.5f * (s_co[k][0] * dx * dx + s_co[k][2] * dy * dy) - s_co[k][1] * dx * dy;
            if (power > 0.0f) continue;
            const float alpha = fminf(0.99f, s_co[k][3] * expf(power));
            if (alpha < 1.0f / 255.0f) continue;
            const float tT = T * (1.0f - alpha);
            if (tT < 1e-4f) { done = true; break; }
            const float w = alpha * T;
            C0 += s_rgb[k][0] * w; C1 += s_rgb[k][1] * w; C2 += s_rgb[k][2] * w; Wt += w;
            T = tT;
        }
        if (!done) s_done = 1;                 // somebody still needs more Gaussians
        __syncthreads();
        if (s_done == 0) break;
        __syncthreads();
    }
    if (inside) {
        const long hw = (long)size * size, o = (long)py * size + px;
        const float r = C0 + T * bg[0], g = C1 + T * bg[1], b = C2 + T * bg[2];
        out_color[o] = fminf(1.f, fmaxf(0.f, r));              // core/gs.py:84  rendered_image.clamp(0, 1)
        out_color[hw + o] = fminf(1.f, fmaxf(0.f, g));
        out_color[2 * hw + o] = fminf(1.f, fmaxf(0.f, b));
        if (out_alpha) out_alpha[o] = Wt;
    }
}

__global__ __launch_bounds__(256) void gs_render_kernel(const VmvGsParams p) {
    const int grid = (p.size + GS_TILE - 1) / GS_TILE;
    const int tile = blockIdx.y * grid + blockIdx.x;
    gs_blend_tile(p.size, p.bg, p.ranges[2 * tile], p.ranges[2 * tile + 1], p.vals_sorted, p.xy, p.conic_opacity, p.gaussians,
                  p.out_color, p.out_alpha);
}

__global__ __launch_bounds__(256) void gs_render_batch_kernel(const VmvGsBatchParams p) {
    const int grid = (p.size + GS_TILE - 1) / GS_TILE;
    const int vv = blockIdx.z;
    const long tile = (long)vv * grid * grid + blockIdx.y * grid + blockIdx.x;
    const long hw = (long)p.size * p.size, vo = (long)vv * p.N;
    // (a view without instances has ranges 0 / 0 from the memset: the tile writes the background)
    gs_blend_tile(p.size, p.bg, p.ranges[2 * tile], p.ranges[2 * tile + 1], p.vals_sorted, p.xy + 2 * vo, p.conic_opacity + 4 * vo,
                  p.gaussians + (long)(vv / p.V) * p.N * 14, p.out_color + 3 * hw * vv, p.out_alpha ? p.out_alpha + hw * vv : nullptr);
}

int gs_check(const VmvGsParams& p) {
    if (!p.gaussians || !p.view || !p.view_proj || !p.depth || !p.xy || !p.conic_opacity || !p.rect || !p.tiles_touched ||
        !p.offsets)
        return VMV_ENULL;
    if (p.N <= 0 || p.size <= 0 || !(p.tan_half_fov > 0.f)) return VMV_EINVAL;
    return VMV_OK;
}

}  // namespace

extern "C" int vmv_gs_workspace_bytes(int n_gaussians, int n_instances, size_t* scan_bytes, size_t* sort_bytes) {
    if (!scan_bytes || !sort_bytes || n_gaussians <= 0 || n_instances < 0) return VMV_EINVAL;
    uint32_t* a = nullptr;
    uint64_t* k = nullptr;
    hipError_t e = rocprim::inclusive_scan(nullptr, *scan_bytes, a, a, (size_t)n_gaussians, rocprim::plus<uint32_t>(), (hipStream_t)0);
    if (e != hipSuccess) return (int)e;
    e = rocprim::radix_sort_pairs(nullptr, *sort_bytes, k, k, a, a, (size_t)(n_instances > 0 ? n_instances : 1), 0u, 64u,
                                  (hipStream_t)0);
    return e == hipSuccess ? VMV_OK : (int)e;
}

extern "C" int vmv_gs_preprocess(const VmvGsParams* pp, void* stream) {
    if (!pp) return VMV_ENULL;
    const VmvGsParams& p = *pp;
    int rc = gs_check(p);
    if (rc != VMV_OK) return rc;
    if (!p.scan_temp) return VMV_ENULL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(gs_preprocess_kernel, dim3((p.N + 255) / 256), dim3(256), 0, st, p);
    size_t bytes = p.scan_temp_bytes;
    hipError_t e = rocprim::inclusive_scan(p.scan_temp, bytes, p.tiles_touched, p.offsets, (size_t)p.N, rocprim::plus<uint32_t>(), st);
    if (e != hipSuccess) return (int)e;
    return vmv_launch_status();
}

extern "C" int vmv_gs_render(const VmvGsParams* pp, void* stream) {
    if (!pp) return VMV_ENULL;
    const VmvGsParams& p = *pp;
    int rc = gs_check(p);
    if (rc != VMV_OK) return rc;
    if (!p.ranges || !p.out_color) return VMV_ENULL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int grid = (p.size + GS_TILE - 1) / GS_TILE;
    hipError_t e = hipMemsetAsync(p.ranges, 0, sizeof(uint32_t) * 2 * grid * grid, st);
    if (e != hipSuccess) return (int)e;
    if (p.num_rendered > 0) {
        if (!p.keys || !p.keys_sorted || !p.vals || !p.vals_sorted || !p.sort_temp) return VMV_ENULL;
        hipLaunchKernelGGL(gs_duplicate_kernel, dim3((p.N + 255) / 256), dim3(256), 0, st, p);
        int tile_bits = 1;
        while ((1 << tile_bits) < grid * grid) ++tile_bits;
        size_t bytes = p.sort_temp_bytes;
        e = rocprim::radix_sort_pairs(p.sort_temp, bytes, p.keys, p.keys_sorted, p.vals, p.vals_sorted, (size_t)p.num_rendered,
                                      0u, (unsigned)(32 + tile_bits), st);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(gs_ranges_kernel, dim3((p.num_rendered + 255) / 256), dim3(256), 0, st, p);
    }
    hipLaunchKernelGGL(gs_render_kernel, dim3(grid, grid), dim3(256), 0, st, p);
    return vmv_launch_status();
}

// ---- batched entry points (vmv.h VmvGsBatchParams)
namespace {
int gs_batch_check(const VmvGsBatchParams& p) {
    if (!p.gaussians || !p.views || !p.view_projs || !p.depth || !p.xy || !p.conic_opacity || !p.rect || !p.tiles_touched || !p.offsets)
        return VMV_ENULL;
    if (p.B <= 0 || p.V <= 0 || p.N <= 0 || p.size <= 0 || !(p.tan_half_fov > 0.f)) return VMV_EINVAL;
    const long grid = (p.size + GS_TILE - 1) / GS_TILE;
    if ((long)p.B * p.V > 65535 || (long)p.B * p.V * p.N >= (1L << 31) || (long)p.B * p.V * grid * grid >= (1L << 31)) return VMV_ERANGE;
    return VMV_OK;
}
}  // namespace

extern "C" int vmv_gs_batch_workspace_bytes(int n_view_gaussians, int n_instances, int key_bits, size_t* scan_bytes, size_t* sort_bytes) {
    if (!scan_bytes || !sort_bytes || n_view_gaussians <= 0 || n_instances < 0 || key_bits < 1 || key_bits > 64) return VMV_EINVAL;
    uint32_t* a = nullptr;
    uint64_t* k = nullptr;
    hipError_t e = rocprim::inclusive_scan(nullptr, *scan_bytes, a, a, (size_t)n_view_gaussians, rocprim::plus<uint32_t>(), (hipStream_t)0);
    if (e != hipSuccess) return (int)e;
    e = rocprim::radix_sort_pairs(nullptr, *sort_bytes, k, k, a, a, (size_t)(n_instances > 0 ? n_instances : 1), 0u, (unsigned)key_bits,
                                  (hipStream_t)0);
    return e == hipSuccess ? VMV_OK : (int)e;
}

extern "C" int vmv_gs_batch_key_bits(int n_views, int size) {
    if (n_views <= 0 || size <= 0) return VMV_EINVAL;
    const long grid = (size + GS_TILE - 1) / GS_TILE;
    int bits = 1;
    while ((1L << bits) < (long)n_views * grid * grid) ++bits;
    return 32 + bits;
}

extern "C" int vmv_gs_batch_preprocess(const VmvGsBatchParams* pp, void* stream) {
    if (!pp) return VMV_ENULL;
    const VmvGsBatchParams& p = *pp;
    int rc = gs_batch_check(p);
    if (rc != VMV_OK) return rc;
    if (!p.scan_temp) return VMV_ENULL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(gs_preprocess_batch_kernel, dim3((p.N + 255) / 256, p.B * p.V), dim3(256), 0, st, p);
    size_t bytes = p.scan_temp_bytes;
    hipError_t e = rocprim::inclusive_scan(p.scan_temp, bytes, p.tiles_touched, p.offsets, (size_t)p.B * p.V * p.N, rocprim::plus<uint32_t>(), st);
    if (e != hipSuccess) return (int)e;
    return vmv_launch_status();
}

extern "C" int vmv_gs_batch_render(const VmvGsBatchParams* pp, void* stream) {
    if (!pp) return VMV_ENULL;
    const VmvGsBatchParams& p = *pp;
    int rc = gs_batch_check(p);
    if (rc != VMV_OK) return rc;
    if (!p.ranges || !p.out_color) return VMV_ENULL;
    if (p.num_rendered < 0) return VMV_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int grid = (p.size + GS_TILE - 1) / GS_TILE;
    const int VV = p.B * p.V;
    hipError_t e = hipMemsetAsync(p.ranges, 0, sizeof(uint32_t) * 2 * (size_t)VV * grid * grid, st);
    if (e != hipSuccess) return (int)e;
    if (p.num_rendered > 0) {
        if (!p.keys || !p.keys_sorted || !p.vals || !p.vals_sorted || !p.sort_temp) return VMV_ENULL;
        hipLaunchKernelGGL(gs_duplicate_batch_kernel, dim3((p.N + 255) / 256, VV), dim3(256), 0, st, p);
        size_t bytes = p.sort_temp_bytes;
        e = rocprim::radix_sort_pairs(p.sort_temp, bytes, p.keys, p.keys_sorted, p.vals, p.vals_sorted, (size_t)p.num_rendered,
                                      0u, (unsigned)vmv_gs_batch_key_bits(VV, p.size), st);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(gs_ranges_batch_kernel, dim3((p.num_rendered + 255) / 256), dim3(256), 0, st, p);
    }
    hipLaunchKernelGGL(gs_render_batch_kernel, dim3(grid, grid, VV), dim3(256), 0, st, p);
    return vmv_launch_status();
}

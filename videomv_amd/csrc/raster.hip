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
//
// The BATCHED pass (vmv_gs_batch_*, round 6) keeps the depth out of the big sort.  The V x N (view, Gaussian) records are first sorted by
// (view, depth bits) — a 64-bit-key radix sort of 1.6 M pairs, stable, so equal depths stay in Gaussian-index order — and the instances
// are emitted in THAT order; the instance sort then runs on the 32-bit (view, tile) id alone — two 8-bit radix passes over 8-byte pairs
// for 24 views x 1024 tiles instead of six over 12-byte pairs — and, being stable, leaves every tile's instances ordered by (depth,
// Gaussian index): exactly the order of the per-view path's 64-bit (tile, depth) stable sort (a tile holds a Gaussian at most once),
// so both paths blend the same sequence and give the same bits (tests/test_gs_gpu.py::test_batched_pass_equals_the_per_view_loop).
// The duplicate pass is block-cooperative: the offsets of 256 depth-ranked Gaussians in LDS, one thread per INSTANCE slot (owner by
// binary search), so the key / value stores are coalesced (the one-thread-per-Gaussian loop wrote ~20 scattered runs per thread:
// 0.67 ms for 32 M instances, now 0.10).  Measured and rejected on the way: the tile-id sort followed by a per-tile bitonic sort of
// (depth, index) keys in the blend block's LDS — correct, but 2.8 ms of sorting for 24 576 tiles (profiles/r6_gs_*).
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

// Preprocess workspace of the batched pass, carved out of VmvGsBatchParams.scan_temp (vmv_gs_batch_workspace_bytes sizes it): the
// depth-ranking sort's key / value buffers, the tile counts in rank order, rocprim's temporaries.  `perm[r]` = index o = vv * N + i of the
// (view, Gaussian) record with rank r in (view, depth, Gaussian index) order.
struct GsBatchWs {
    uint64_t* dkeys; uint64_t* dkeys_sorted;
    uint32_t* perm_in; uint32_t* perm;
    uint32_t* touched_ranked;
    void* prim; size_t prim_bytes;
};
inline size_t gs_align(size_t v) { return (v + 255) & ~(size_t)255; }
inline size_t gs_batch_ws_fixed_bytes(size_t vn) { return 2 * gs_align(vn * 8) + 3 * gs_align(vn * 4); }
inline GsBatchWs gs_batch_ws(void* base, size_t bytes, size_t vn) {
    unsigned char* q = reinterpret_cast<unsigned char*>(base);
    GsBatchWs w;
    w.dkeys = reinterpret_cast<uint64_t*>(q); q += gs_align(vn * 8);
    w.dkeys_sorted = reinterpret_cast<uint64_t*>(q); q += gs_align(vn * 8);
    w.perm_in = reinterpret_cast<uint32_t*>(q); q += gs_align(vn * 4);
    w.perm = reinterpret_cast<uint32_t*>(q); q += gs_align(vn * 4);
    w.touched_ranked = reinterpret_cast<uint32_t*>(q); q += gs_align(vn * 4);
    w.prim = q;
    const size_t fixed = gs_batch_ws_fixed_bytes(vn);
    w.prim_bytes = bytes > fixed ? bytes - fixed : 0;
    return w;
}

// key of the depth-ranking sort: (view << 32) | depth bits (depth > 0.2: the bit pattern orders like the float); records that touch no
// tile sort last in their view (their depth was never written)
__global__ __launch_bounds__(256) void gs_depth_keys_batch_kernel(const VmvGsBatchParams p, uint64_t* __restrict__ dkeys, uint32_t* __restrict__ perm_in) {
    const long o = (long)blockIdx.x * 256 + threadIdx.x;
    if (o >= (long)p.B * p.V * p.N) return;
    const uint32_t vv = (uint32_t)(o / p.N);
    const uint32_t d = p.tiles_touched[o] ? __float_as_uint(p.depth[o]) : 0xffffffffu;
    dkeys[o] = ((uint64_t)vv << 32) | d;
    perm_in[o] = (uint32_t)o;
}
__global__ __launch_bounds__(256) void gs_rank_counts_batch_kernel(const VmvGsBatchParams p, const uint32_t* __restrict__ perm, uint32_t* __restrict__ touched_ranked) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= (long)p.B * p.V * p.N) return;
    touched_ranked[r] = p.tiles_touched[perm[r]];
}

// Block-cooperative duplicate: the block owns 256 consecutive RANKS (records in (view, depth) order); their inclusive offsets and
// rectangles sit in LDS, and thread j of every 256-slot round writes instance slot j of the block's contiguous output range — its owner
// found by binary search over the LDS offsets — so keys and values are stored coalesced.  Key = the 32-bit (view, tile) id; instance k of
// a Gaussian is tile (y0 + k / w, x0 + k % w), the enumeration order of the per-view kernel.
__global__ __launch_bounds__(256) void gs_duplicate_batch_kernel(const VmvGsBatchParams p, const uint32_t* __restrict__ perm) {
    __shared__ uint32_t s_off[256];          // inclusive offsets relative to the block's base
    __shared__ uint32_t s_tile[256];         // (view, tile) id of the rectangle's first tile
    __shared__ int s_w[256];                 // rectangle width in tiles
    __shared__ uint32_t s_gi[256];           // Gaussian index inside its sample
    const long total = (long)p.B * p.V * p.N;
    const long r0 = (long)blockIdx.x * 256, r = r0 + threadIdx.x;
    const uint32_t base = r0 == 0 ? 0u : p.offsets[r0 - 1];
    const int grid = (p.size + GS_TILE - 1) / GS_TILE;
    uint32_t off_incl = 0;
    if (r < total) {
        off_incl = p.offsets[r];
        const long o = perm[r];
        const int* rc = p.rect + 4L * o;
        const int vv = (int)(o / p.N);
        const bool any = p.tiles_touched[o] != 0;
        s_tile[threadIdx.x] = any ? (uint32_t)vv * (uint32_t)(grid * grid) + (uint32_t)(rc[1] * grid + rc[0]) : 0u;
        s_w[threadIdx.x] = any ? rc[2] - rc[0] : 1;
        s_gi[threadIdx.x] = (uint32_t)(o - (long)vv * p.N);
    }
    const long last = (r0 + 255 < total ? r0 + 255 : total - 1);
    s_off[threadIdx.x] = (r < total ? off_incl : p.offsets[last]) - base;
    __syncthreads();
    const uint32_t nblk = s_off[255];
    const uint32_t lim = (uint32_t)p.num_rendered;
    uint32_t* keys32 = reinterpret_cast<uint32_t*>(p.keys);
    for (uint32_t j = threadIdx.x; j < nblk; j += 256) {
        int lo = 0, hi = 255;                // smallest t with s_off[t] > j
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_off[mid] > j) hi = mid; else lo = mid + 1; }
        const uint32_t k = j - (lo == 0 ? 0u : s_off[lo - 1]);
        const int w = s_w[lo];
        const uint32_t ky = k / (uint32_t)w, kx = k - ky * (uint32_t)w;
        const uint32_t dst = base + j;
        if (dst >= lim) break;
        keys32[dst] = s_tile[lo] + ky * (uint32_t)grid + kx;
        p.vals[dst] = s_gi[lo];
    }
}

__global__ __launch_bounds__(256) void gs_ranges_batch_kernel(const VmvGsBatchParams p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.num_rendered) return;
    const uint32_t* ks = reinterpret_cast<const uint32_t*>(p.keys_sorted);
    const uint32_t t = ks[i];
    if (i == 0) p.ranges[2 * t] = 0;
    else {
        const uint32_t tp = ks[i - 1];
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
// `order(j)` = Gaussian index of the tile's j-th instance in blend order, j in [lo, hi).
// Round 6: the inner loop is branch-free per lane and wave-uniform in its control flow.  The divergent form (three nested `continue` /
// `break` tests) spent as many scalar mask instructions as vector ones — 35 + 35 per Gaussian and wave in the ISA — on a kernel that is
// VALU-bound (8 G pixel x Gaussian evaluations per 24 views; it was priced against HBM before, wrongly).  Now the exponent is taken in
// base 2 with the conic pre-scaled by -0.5 log2(e) when the Gaussian is staged (one v_exp_f32 instead of expf's range reduction), a
// WAVE skips a PAIR of Gaussians none of its 64 pixels can see (conservative pre-test on p2 + log2(opacity), before the exponential;
// the pair's exponents are packed fp32 operations), the
// per-lane tests are selects, and the loop leaves when every lane of the wave is done.  Same rule as before per pixel: skip power > 0,
// alpha = min(0.99, opacity e^power), skip alpha < 1/255, stop BEFORE the Gaussian that would take T below 1e-4.
template <typename Order>
VMV_DEV void gs_blend_tile(const int size, const float* __restrict__ bg, const uint32_t lo, const uint32_t hi,
                           const Order order, const float* __restrict__ xy, const float* __restrict__ conic_opacity,
                           const float* __restrict__ gaussians, float* __restrict__ out_color, float* __restrict__ out_alpha) {
    // staged Gaussians, one array per field (+ 2 pad entries): the loop takes them TWO at a time, a field pair is one ds_read_b64 and
    // the falloff exponents of the pair are packed fp32 operations (v_pk_*: 8 for two Gaussians where the one-at-a-time form issued 16)
    __shared__ __attribute__((aligned(16))) float s_x[258], s_y[258], s_A[258], s_B[258], s_C[258], s_l[258], s_o[258], s_r[258], s_g[258], s_b[258];
    __shared__ int s_done;
    const int px = blockIdx.x * GS_TILE + (threadIdx.x & 15), py = blockIdx.y * GS_TILE + (threadIdx.x >> 4);
    const bool inside = px < size && py < size;
    const float fx = (float)px, fy = (float)py;
    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Wt = 0.f;
    bool done = !inside;
    constexpr float LOG2E = 1.4426950408889634f;
    constexpr float CULL = -7.9943534f - 0.02f;      // log2(1 / 255) with a margin: the exact alpha test follows for what passes
    for (uint32_t base = lo; base < hi; base += 256) {
        if (threadIdx.x == 0) s_done = 0;
        __syncthreads();
        const uint32_t j = base + threadIdx.x;
        if (j < hi) {
            const uint32_t gi = order(j);
            const float* co = conic_opacity + 4L * gi;
            const float* g = gaussians + 14L * gi + 11;
            const float o = co[3];
            s_x[threadIdx.x] = xy[2 * gi]; s_y[threadIdx.x] = xy[2 * gi + 1];
            s_A[threadIdx.x] = -0.5f * LOG2E * co[0]; s_B[threadIdx.x] = -LOG2E * co[1]; s_C[threadIdx.x] = -0.5f * LOG2E * co[2];
            s_l[threadIdx.x] = o > 0.f ? __builtin_amdgcn_logf(o) : -1e30f;      // v_log_f32: log2
            s_o[threadIdx.x] = o;
            s_r[threadIdx.x] = g[0]; s_g[threadIdx.x] = g[1]; s_b[threadIdx.x] = g[2];
        } else {                               // pad: a Gaussian nobody sees (the pair of an odd tail)
            s_x[threadIdx.x] = 0.f; s_y[threadIdx.x] = 0.f; s_A[threadIdx.x] = 0.f; s_B[threadIdx.x] = 0.f; s_C[threadIdx.x] = 0.f;
            s_l[threadIdx.x] = -1e30f; s_o[threadIdx.x] = 0.f;
            s_r[threadIdx.x] = 0.f; s_g[threadIdx.x] = 0.f; s_b[threadIdx.x] = 0.f;      // (weight 0 x stale LDS could be 0 x NaN)
        }
        __syncthreads();
        const int n = (int)min(256u, hi - base);
        // one Gaussian against this lane's pixel (p2 = log2 of its falloff there, already known to pass the pre-test in some lane)
        auto blend_one = [&](const int k, const float p2, const bool maybe) {
            const float alpha = fminf(0.99f, s_o[k] * __builtin_amdgcn_exp2f(p2));
            const bool valid = maybe && !done && alpha >= 1.0f / 255.0f;
            const float tT = T * (1.0f - alpha);
            const bool stop = valid && tT < 1e-4f;
            done = done || stop;
            const bool apply = valid && !stop;
            const float w = apply ? alpha * T : 0.0f;
            C0 += s_r[k] * w; C1 += s_g[k] * w; C2 += s_b[k] * w; Wt += w;
            T = apply ? tT : T;
        };
        const f32x2_t fx2 = {fx, fx}, fy2 = {fy, fy};
        for (int k = 0; k < n; k += 2) {       // (entry n of an odd n is a pad or, in a full batch, never read: n = 256 is even)
            const f32x2_t dx = *reinterpret_cast<const f32x2_t*>(s_x + k) - fx2, dy = *reinterpret_cast<const f32x2_t*>(s_y + k) - fy2;
            const f32x2_t p2 = *reinterpret_cast<const f32x2_t*>(s_A + k) * dx * dx + *reinterpret_cast<const f32x2_t*>(s_C + k) * dy * dy +
                               *reinterpret_cast<const f32x2_t*>(s_B + k) * dx * dy;
            const f32x2_t pre = p2 + *reinterpret_cast<const f32x2_t*>(s_l + k);
            const bool m0 = p2.x <= 0.0f && pre.x >= CULL, m1 = p2.y <= 0.0f && pre.y >= CULL;
            if (__builtin_amdgcn_ballot_w64(!done && (m0 || m1)) == 0) continue;      // nobody in the wave sees either
            blend_one(k, p2.x, m0);
            blend_one(k + 1, p2.y, m1);
            if (__builtin_amdgcn_ballot_w64(!done) == 0) break;                        // the whole wave is saturated
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
    const uint32_t* vs = p.vals_sorted;
    gs_blend_tile(p.size, p.bg, p.ranges[2 * tile], p.ranges[2 * tile + 1], [vs](const uint32_t j) { return vs[j]; }, p.xy, p.conic_opacity,
                  p.gaussians, p.out_color, p.out_alpha);
}

__global__ __launch_bounds__(256) void gs_render_batch_kernel(const VmvGsBatchParams p) {
    const int grid = (p.size + GS_TILE - 1) / GS_TILE;
    const int vv = blockIdx.z;
    const long tile = (long)vv * grid * grid + blockIdx.y * grid + blockIdx.x;
    const long hw = (long)p.size * p.size, vo = (long)vv * p.N;
    // (a view without instances has ranges 0 / 0 from the memset: the tile writes the background)
    const uint32_t* vs = p.vals_sorted;
    gs_blend_tile(p.size, p.bg, p.ranges[2 * tile], p.ranges[2 * tile + 1], [vs](const uint32_t j) { return vs[j]; }, p.xy + 2 * vo,
                  p.conic_opacity + 4 * vo, p.gaussians + (long)(vv / p.V) * p.N * 14, p.out_color + 3 * hw * vv,
                  p.out_alpha ? p.out_alpha + hw * vv : nullptr);
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
    // scan_temp = the preprocess workspace (GsBatchWs): the depth-ranking sort's buffers + the larger of the two primitives' temporaries
    size_t scan_prim = 0, rank_prim = 0;
    hipError_t e = rocprim::inclusive_scan(nullptr, scan_prim, a, a, (size_t)n_view_gaussians, rocprim::plus<uint32_t>(), (hipStream_t)0);
    if (e != hipSuccess) return (int)e;
    e = rocprim::radix_sort_pairs(nullptr, rank_prim, k, k, a, a, (size_t)n_view_gaussians, 0u, 64u, (hipStream_t)0);
    if (e != hipSuccess) return (int)e;
    *scan_bytes = gs_batch_ws_fixed_bytes((size_t)n_view_gaussians) + gs_align(scan_prim > rank_prim ? scan_prim : rank_prim);
    e = rocprim::radix_sort_pairs(nullptr, *sort_bytes, a, a, a, a, (size_t)(n_instances > 0 ? n_instances : 1), 0u,
                                  (unsigned)(key_bits > 32 ? key_bits - 32 : key_bits), (hipStream_t)0);      // 32-bit (view, tile) keys
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
    const size_t vn = (size_t)p.B * p.V * p.N;
    if (!vmv_aligned16(p.scan_temp)) return VMV_EALIGN;
    if (p.scan_temp_bytes <= gs_batch_ws_fixed_bytes(vn)) return VMV_ERANGE;
    const GsBatchWs w = gs_batch_ws(p.scan_temp, p.scan_temp_bytes, vn);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const unsigned nb = (unsigned)((vn + 255) / 256);
    hipLaunchKernelGGL(gs_preprocess_batch_kernel, dim3((p.N + 255) / 256, p.B * p.V), dim3(256), 0, st, p);
    // rank the records of every view by depth (stable: equal depths keep their Gaussian-index order), count tiles in rank order, scan
    hipLaunchKernelGGL(gs_depth_keys_batch_kernel, dim3(nb), dim3(256), 0, st, p, w.dkeys, w.perm_in);
    int vbits = 1;
    while ((1L << vbits) < (long)p.B * p.V) ++vbits;
    size_t bytes = w.prim_bytes;
    hipError_t e = rocprim::radix_sort_pairs(w.prim, bytes, w.dkeys, w.dkeys_sorted, w.perm_in, w.perm, vn, 0u, (unsigned)(32 + vbits), st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(gs_rank_counts_batch_kernel, dim3(nb), dim3(256), 0, st, p, w.perm, w.touched_ranked);
    bytes = w.prim_bytes;
    e = rocprim::inclusive_scan(w.prim, bytes, w.touched_ranked, p.offsets, vn, rocprim::plus<uint32_t>(), st);
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
        if (!p.scan_temp || p.scan_temp_bytes <= gs_batch_ws_fixed_bytes((size_t)VV * p.N)) return VMV_ENULL;      // perm[] of _preprocess
        const GsBatchWs w = gs_batch_ws(p.scan_temp, p.scan_temp_bytes, (size_t)VV * p.N);
        hipLaunchKernelGGL(gs_duplicate_batch_kernel, dim3((unsigned)(((long)VV * p.N + 255) / 256)), dim3(256), 0, st, p, (const uint32_t*)w.perm);
        size_t bytes = p.sort_temp_bytes;
        // (view, tile) ids only: keys are the first 4 n bytes of the 8 n-byte key buffers (header comment)
        e = rocprim::radix_sort_pairs(p.sort_temp, bytes, reinterpret_cast<uint32_t*>(p.keys), reinterpret_cast<uint32_t*>(p.keys_sorted),
                                      p.vals, p.vals_sorted, (size_t)p.num_rendered, 0u, (unsigned)(vmv_gs_batch_key_bits(VV, p.size) - 32), st);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(gs_ranges_batch_kernel, dim3((p.num_rendered + 255) / 256), dim3(256), 0, st, p);
    }
    hipLaunchKernelGGL(gs_render_batch_kernel, dim3(grid, grid, VV), dim3(256), 0, st, p);
    return vmv_launch_status();
}

// Micro-benchmark (experiment, not product): how well do the three streams of the GEMM main loop overlap on one CU when
// NOTHING synchronises the waves?  Per "chunk" a wave issues 6 LDS-DMA wave-instructions (L2-resident source), 16
// ds_read_b128 fragment reads and 32 MFMA 16x16x32 bf16 — the 256x128x64 tile's per-wave work — in every on/off combination.
// 8 waves per CU (512-thread blocks, 256 blocks).   hipcc --offload-arch=gfx950 -O3 cu_overlap.hip -o cu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

__device__ inline void blds16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lptr, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lptr), 16, voff, soff, 0, 0);
}

template <bool DMA, bool RD, bool MM, bool BAR>
__global__ __launch_bounds__(512) void k(const unsigned char* src, uint32_t window_groups, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), 0, 0x7ffffff0u, 0x00020000u);
    const uint32_t lane_off = (uint32_t)((lane >> 3) * 640 + (lane & 7) * 16);
    uint32_t g = (blockIdx.x * 8u + wave) * 2654435761u;
    unsigned char* ring = smem + 65536 + wave * 12 * 1024;          // 2 chunks x 6 KB per wave
    const int frow = lane & 15, fgrp = lane >> 4, fswz = (frow >> 1) & 7;
    const u32x4_t* fa = reinterpret_cast<const u32x4_t*>(smem) + ((wave >> 1) * 64 + frow) * 8;
    const u32x4_t* fw = reinterpret_cast<const u32x4_t*>(smem + 32768) + ((wave & 1) * 64 + frow) * 8;
    f32x4_t acc[4][4];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0, 0, 0, 0};
    bf16x8_t a[2][4], w[2][4];
    for (int b = 0; b < 2; ++b) for (int i = 0; i < 4; ++i) { a[b][i] = __builtin_bit_cast(bf16x8_t, fa[i * 128 + lane]); w[b][i] = __builtin_bit_cast(bf16x8_t, fw[i * 128 + lane]); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int slot = (kk * 4 + fgrp) ^ fswz;
            if constexpr (RD) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { a[kk ^ 1][i] = __builtin_bit_cast(bf16x8_t, fa[i * 128 + slot]); w[kk ^ 1][i] = __builtin_bit_cast(bf16x8_t, fw[i * 128 + slot]); }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MM) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[kk][j], a[kk][i], acc[j][i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kk == 0 && BAR) { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier(); }
        }
        if constexpr (DMA) {
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                g = g * 1664525u + 1013904223u;
                const uint32_t soff = __builtin_amdgcn_readfirstlane(((g >> 8) % window_groups) * 5120u);
                blds16(rsrc, ring + ((it & 1) * 6 + q) * 1024, lane_off, soff);
            }
            if (!BAR) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0;
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) s += acc[j][i].x + acc[j][i].y + acc[j][i].z + acc[j][i].w;
    if (s == 12345.f) sink[threadIdx.x] = s;
}

// V1: the 6 DMA pieces spread between the MFMAs (one per ~5 MFMAs); V3/V4: NL dedicated loader waves next to the 8 MFMA waves
template <bool RD, bool BAR, int EVERY>
__global__ __launch_bounds__(512) void k_il(const unsigned char* src, uint32_t window_groups, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), 0, 0x7ffffff0u, 0x00020000u);
    const uint32_t lane_off = (uint32_t)((lane >> 3) * 640 + (lane & 7) * 16);
    uint32_t g = (blockIdx.x * 8u + wave) * 2654435761u;
    unsigned char* ring = smem + 65536 + wave * 12 * 1024;
    const int frow = lane & 15, fgrp = lane >> 4, fswz = (frow >> 1) & 7;
    const u32x4_t* fa = reinterpret_cast<const u32x4_t*>(smem) + ((wave >> 1) * 64 + frow) * 8;
    const u32x4_t* fw = reinterpret_cast<const u32x4_t*>(smem + 32768) + ((wave & 1) * 64 + frow) * 8;
    f32x4_t acc[4][4];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0, 0, 0, 0};
    bf16x8_t a[2][4], w[2][4];
    for (int b = 0; b < 2; ++b) for (int i = 0; i < 4; ++i) { a[b][i] = __builtin_bit_cast(bf16x8_t, fa[i * 128 + lane]); w[b][i] = __builtin_bit_cast(bf16x8_t, fw[i * 128 + lane]); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int slot = (kk * 4 + fgrp) ^ fswz;
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const int j = m >> 2, i = m & 3;
                acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[kk][j], a[kk][i], acc[j][i], 0, 0, 0);
                if (RD && (m & 1)) {
                    const int r = m >> 1;
                    if (r < 4) a[kk ^ 1][r & 3] = __builtin_bit_cast(bf16x8_t, fa[(r & 3) * 128 + slot]);
                    else w[kk ^ 1][r & 3] = __builtin_bit_cast(bf16x8_t, fw[(r & 3) * 128 + slot]);
                }
                if ((kk * 16 + m) % EVERY == EVERY - 1 && (kk * 16 + m) / EVERY < 6) {
                    const int q = (kk * 16 + m) / EVERY;
                    g = g * 1664525u + 1013904223u;
                    const uint32_t soff = __builtin_amdgcn_readfirstlane(((g >> 8) % window_groups) * 5120u);
                    blds16(rsrc, ring + ((it & 1) * 6 + q) * 1024, lane_off, soff);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (kk == 0 && BAR) { asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier(); }
        }
        if (!BAR) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0;
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) s += acc[j][i].x + acc[j][i].y + acc[j][i].z + acc[j][i].w;
    if (s == 12345.f) sink[threadIdx.x] = s;
}

template <int NL, bool RD, bool BAR>
__global__ __launch_bounds__(512 + 64 * NL) void k_spec(const unsigned char* src, uint32_t window_groups, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= 8) {                      // loader waves: 48 / NL pieces per chunk each, two chunks in flight
        constexpr int PER = 48 / NL;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), 0, 0x7ffffff0u, 0x00020000u);
        const uint32_t lane_off = (uint32_t)((lane >> 3) * 640 + (lane & 7) * 16);
        uint32_t g = (blockIdx.x * 8u + wave) * 2654435761u;
        unsigned char* ring = smem + 65536 + (wave - 8) * 2 * PER * 1024;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                g = g * 1664525u + 1013904223u;
                const uint32_t soff = __builtin_amdgcn_readfirstlane(((g >> 8) % window_groups) * 5120u);
                blds16(rsrc, ring + ((it & 1) * PER + q) * 1024, lane_off, soff);
            }
            if (PER == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            if (BAR) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    const int frow = lane & 15, fgrp = lane >> 4, fswz = (frow >> 1) & 7;
    const u32x4_t* fa = reinterpret_cast<const u32x4_t*>(smem) + ((wave >> 1) * 64 + frow) * 8;
    const u32x4_t* fw = reinterpret_cast<const u32x4_t*>(smem + 32768) + ((wave & 1) * 64 + frow) * 8;
    f32x4_t acc[4][4];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{0, 0, 0, 0};
    bf16x8_t a[2][4], w[2][4];
    for (int b = 0; b < 2; ++b) for (int i = 0; i < 4; ++i) { a[b][i] = __builtin_bit_cast(bf16x8_t, fa[i * 128 + lane]); w[b][i] = __builtin_bit_cast(bf16x8_t, fw[i * 128 + lane]); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int slot = (kk * 4 + fgrp) ^ fswz;
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const int j = m >> 2, i = m & 3;
                acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[kk][j], a[kk][i], acc[j][i], 0, 0, 0);
                if (RD && (m & 1)) {
                    const int r = m >> 1;
                    if (r < 4) a[kk ^ 1][r & 3] = __builtin_bit_cast(bf16x8_t, fa[(r & 3) * 128 + slot]);
                    else w[kk ^ 1][r & 3] = __builtin_bit_cast(bf16x8_t, fw[(r & 3) * 128 + slot]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (BAR) { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier(); }
        }
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) s += acc[j][i].x + acc[j][i].y + acc[j][i].z + acc[j][i].w;
    if (s == 12345.f) sink[threadIdx.x] = s;
}

template <class K>
void run_k(const char* name, K kern, int threads, const unsigned char* src, float* sink) {
    const int iters = 2000;
    const uint32_t wg = (1u << 20) / 5120u;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 160 * 1024, 0, src, wg, iters, sink);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 160 * 1024, 0, src, wg, iters, sink);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ns = ms * 1e6 / iters;
    printf("%-44s %7.1f ns per chunk   (MFMA-equivalent %6.0f TFLOP/s chip)\n", name, ns, 256.0 * 256 * 128 * 64 * 2 / ns / 1e3);
}

template <bool DMA, bool RD, bool MM, bool BAR>
void run(const char* name, const unsigned char* src, float* sink) {
    const int iters = 2000;
    const uint32_t wg = (1u << 20) / 5120u;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<DMA, RD, MM, BAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((k<DMA, RD, MM, BAR>), dim3(256), dim3(512), 160 * 1024, 0, src, wg, iters, sink);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<DMA, RD, MM, BAR>), dim3(256), dim3(512), 160 * 1024, 0, src, wg, iters, sink);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ns = ms * 1e6 / iters;
    printf("%-28s %7.1f ns per chunk   (MFMA-equivalent %6.0f TFLOP/s chip, DMA-equivalent %5.1f GB/s per CU)\n", name, ns,
           256.0 * 256 * 128 * 64 * 2 / ns / 1e3, 48.0 * 1024 / ns);
}

int main() {
    unsigned char* src; float* sink;
    CHECK(hipMalloc(&src, 64 << 20)); CHECK(hipMemset(src, 0, 64 << 20)); CHECK(hipMalloc(&sink, 4096));
    run<true, false, false, false>("DMA", src, sink);
    run<false, true, false, false>("reads", src, sink);
    run<false, false, true, false>("MFMA", src, sink);
    run<true, true, false, false>("DMA + reads", src, sink);
    run<true, false, true, false>("DMA + MFMA", src, sink);
    run<false, true, true, false>("reads + MFMA", src, sink);
    run<true, true, true, false>("DMA + reads + MFMA", src, sink);
    run<true, true, true, true>("DMA + reads + MFMA + barrier", src, sink);
    run<false, true, true, true>("reads + MFMA + barrier", src, sink);
    run_k("IL every 5: DMA + MFMA", &k_il<false, false, 5>, 512, src, sink);
    run_k("IL every 5: DMA + reads + MFMA", &k_il<true, false, 5>, 512, src, sink);
    run_k("IL every 5: DMA + reads + MFMA + barrier", &k_il<true, true, 5>, 512, src, sink);
    run_k("IL every 2 (front-loaded): all", &k_il<true, false, 2>, 512, src, sink);
    run_k("spec 8 MFMA + 4 loaders: DMA + MFMA", &k_spec<4, false, false>, 768, src, sink);
    run_k("spec 8 MFMA + 4 loaders: all", &k_spec<4, true, false>, 768, src, sink);
    run_k("spec 8 MFMA + 4 loaders: all + barrier", &k_spec<4, true, true>, 768, src, sink);
    run_k("spec 8 MFMA + 8 loaders: DMA + MFMA", &k_spec<8, false, false>, 1024, src, sink);
    run_k("spec 8 MFMA + 8 loaders: all", &k_spec<8, true, false>, 1024, src, sink);
    run_k("spec 8 MFMA + 8 loaders: all + barrier", &k_spec<8, true, true>, 1024, src, sink);
    return 0;
}

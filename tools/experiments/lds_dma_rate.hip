// Micro-benchmark (experiment, not product): sustained L2/MALL/HBM -> LDS rate of 16-byte LDS-DMA (`buffer_load ... lds`)
// per CU as a function of waves per CU and wave-instructions in flight per wave.  Access pattern = the GEMM loader's:
// one wave-instruction moves 8 rows x 128 B (row pitch 640 B).   hipcc --offload-arch=gfx950 -O3 lds_dma_rate.hip -o lds_dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ inline void blds16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lptr, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lptr), 16, voff, soff, 0, 0);
}

template <int D> __device__ inline void wait_le();
#define W(N) template <> __device__ inline void wait_le<N>() { asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); }
W(0) W(1) W(3) W(5) W(7) W(11) W(15) W(23) W(31) W(47)

template <int D>
__global__ __launch_bounds__(1024) void dma_kernel(const unsigned char* src, uint32_t window_groups, int iters, int pitch, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), 0, 0x7ffffff0u, 0x00020000u);
    const uint32_t lane_off = (uint32_t)((lane >> 3) * pitch + (lane & 7) * 16);
    uint32_t g = (blockIdx.x * 8u + wave) * 2654435761u;
    unsigned char* ring = smem + wave * (D < 8 ? D : 8) * 1024;
    for (int it = 0; it < iters; ++it) {
        g = g * 1664525u + 1013904223u;
        const uint32_t grp = (g >> 8) % window_groups;
        const uint32_t soff = __builtin_amdgcn_readfirstlane(grp * (uint32_t)(8 * pitch));
        blds16(rsrc, ring + (it % (D < 8 ? D : 8)) * 1024, lane_off, soff);
        wait_le<D - 1>();
    }
    wait_le<0>();
    __syncthreads();
    if (threadIdx.x == 0 && smem[17] == 0x77) sink[0] = 1;
}

// the same traffic as plain 16-byte global loads into registers (D loads in flight per lane, consumed by an xor)
template <int D>
__global__ __launch_bounds__(1024) void reg_kernel(const unsigned char* src, uint32_t window_groups, int iters, int pitch, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), 0, 0x7ffffff0u, 0x00020000u);
    const uint32_t lane_off = (uint32_t)((lane >> 3) * pitch + (lane & 7) * 16);
    uint32_t g = (blockIdx.x * 16u + wave) * 2654435761u;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; it += D) {
        u4 v[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            g = g * 1664525u + 1013904223u;
            const uint32_t grp = (g >> 8) % window_groups;
            const uint32_t soff = __builtin_amdgcn_readfirstlane(grp * (uint32_t)(8 * pitch));
            v[d] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, soff, 0);
        }
#pragma unroll
        for (int d = 0; d < D; ++d) acc ^= v[d];
    }
    if (acc.x == 0x12345678u) sink[threadIdx.x] = acc.y;
}

template <int D>
float run_reg(const unsigned char* src, uint32_t wg, int nw, int iters, int pitch, unsigned* sink) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(reg_kernel<D>, dim3(256), dim3(64 * nw), 0, 0, src, wg, iters, pitch, sink);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(reg_kernel<D>, dim3(256), dim3(64 * nw), 0, 0, src, wg, iters, pitch, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

template <int D>
float run(const unsigned char* src, uint32_t wg, int nw, int iters, int pitch, unsigned* sink) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int lds = nw * (D < 8 ? D : 8) * 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(dma_kernel<D>, dim3(256), dim3(64 * nw), lds, 0, src, wg, iters, pitch, sink);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(dma_kernel<D>, dim3(256), dim3(64 * nw), lds, 0, src, wg, iters, pitch, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main() {
    const size_t total = 2048ull << 20;
    unsigned char* src; unsigned* sink;
    CHECK(hipMalloc(&src, total)); CHECK(hipMemset(src, 1, total)); CHECK(hipMalloc(&sink, 64));
    int pitch = 640;
    const size_t windows[3] = {1ull << 20, 96ull << 20, 2000ull << 20};
    const char* wname[3] = {"L2 (1 MB)", "MALL (96 MB)", "HBM (2 GB)"};
    for (int wi = 0; wi < 3; ++wi) {
        const uint32_t wg = (uint32_t)(windows[wi] / (8 * pitch));
        for (int nw = 4; nw <= 16; nw *= 2) {
            printf("%-13s waves/CU %2d :", wname[wi], nw);
            const int iters = 4096;
            auto rep = [&](int D, float ms) {
                const double bytes = 256.0 * nw * iters * 1024.0;
                printf("  D%-2d %5.1f GB/s/CU", D, bytes / (ms * 1e-3) / 256 / 1e9);
            };
            rep(2, run<2>(src, wg, nw, iters, pitch, sink));
            rep(4, run<4>(src, wg, nw, iters, pitch, sink));
            rep(6, run<6>(src, wg, nw, iters, pitch, sink));
            rep(12, run<12>(src, wg, nw, iters, pitch, sink));
            rep(16, run<16>(src, wg, nw, iters, pitch, sink));
            rep(24, run<24>(src, wg, nw, iters, pitch, sink));
            rep(32, run<32>(src, wg, nw, iters, pitch, sink));
            rep(48, run<48>(src, wg, nw, iters, pitch, sink));
            printf("\n");
            printf("   -> registers           :");
            rep(2, run_reg<2>(src, wg, nw, iters, pitch, sink));
            rep(4, run_reg<4>(src, wg, nw, iters, pitch, sink));
            rep(8, run_reg<8>(src, wg, nw, iters, pitch, sink));
            rep(16, run_reg<16>(src, wg, nw, iters, pitch, sink));
            printf("\n");
        }
    }
    pitch = 128;
    {
        const uint32_t wg = (uint32_t)((1ull << 20) / (8 * pitch));
        for (int nw = 4; nw <= 16; nw *= 2) {
            printf("L2 contiguous 1 KB/instr, waves/CU %2d :", nw);
            const int iters = 4096;
            auto rep = [&](int D, float ms) { printf("  D%-2d %5.1f GB/s/CU", D, 256.0 * nw * iters * 1024.0 / (ms * 1e-3) / 256 / 1e9); };
            rep(4, run<4>(src, wg, nw, iters, pitch, sink));
            rep(12, run<12>(src, wg, nw, iters, pitch, sink));
            printf("   regs:");
            rep(8, run_reg<8>(src, wg, nw, iters, pitch, sink));
            printf("\n");
        }
    }
    return 0;
}

// Experiment: does an out-of-range lane of `buffer_load_dwordx4 ... lds` write ZEROS to LDS on gfx950?
// (decides whether conv zero-padding can use the buffer bounds check instead of a zero-page pointer select)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
__global__ void k(const uint32_t* src, uint32_t nbytes, uint32_t* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    reinterpret_cast<u32x4_t*>(smem)[lane] = u32x4_t{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    const uint32_t voff = (lane & 1) ? 0x80000000u : (uint32_t)lane * 16u;     // odd lanes out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)smem, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = reinterpret_cast<uint32_t*>(smem)[lane * 4 + i];
}
int main() {
    std::vector<uint32_t> h(256);
    for (int i = 0; i < 256; ++i) h[i] = 0x1000 + i;
    uint32_t *d, *o;
    hipMalloc(&d, 1024); hipMalloc(&o, 1024);
    hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, d, 1024u, o);
    std::vector<uint32_t> r(256);
    hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    int ok_in = 0, zero_oob = 0, stale_oob = 0;
    for (int l = 0; l < 64; ++l) {
        if (l & 1) { if (r[l * 4] == 0) zero_oob++; else if (r[l * 4] == 0xffffffffu) stale_oob++; }
        else if (r[l * 4] == 0x1000u + l * 4) ok_in++;
    }
    printf("in-range ok %d/32, oob lanes: zero %d, stale %d (first oob word %08x)\n", ok_in, zero_oob, stale_oob, r[4]);
    return 0;
}

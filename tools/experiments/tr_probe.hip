// Probe of gfx950's ds_read_b64_tr_b16 semantics (build: hipcc --offload-arch=gfx950 -O2 tr_probe.hip -o tr_probe).
// LDS holds rows of 16 shorts, value = 16 * row + col.  Lane l (j = l & 15, g = l >> 4) reads 8 bytes at row 4 g + j / 4,
// columns 4 (j % 4) .. + 3.  Expected if each 16-lane group transposes its [4 rows][16 cols] block: lane (g, j) receives
// rows 4 g + 0..3 of column j, i.e. 16 (4 g + e) + j for e = 0..3.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short* o) {
    __shared__ short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x, j = l & 15, g = l >> 4;
    s4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + (g * 4 + j / 4) * 16 + 4 * (j % 4)));
    for (int e = 0; e < 4; ++e) o[l * 4 + e] = r[e];
}
int main() {
    short* d; short h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e) bad += h[l * 4 + e] != 16 * (4 * (l >> 4) + e) + (l & 15);
    printf("mismatches vs the [4 rows][16 cols] block-transpose model: %d\n", bad);
    for (int l = 0; l < 64; l += 5) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    return 0;
}

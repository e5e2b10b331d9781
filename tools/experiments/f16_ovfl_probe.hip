// f16_ovfl_probe.hip — does gfx950 honour MODE.FP16_OVFL (bit 23) for v_cvt_pk_f16_f32 / v_cvt_f16_f32?
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/f16_ovfl_probe.hip -o tools/experiments/f16_ovfl_probe && ./tools/experiments/f16_ovfl_probe
// Prints the fp16 bit patterns of {1e5, -1e5, 65504, 65520 (rounds up to inf without the bit), +inf, -inf, NaN, 1.0} converted with and
// without the mode bit.  Expected with the bit: 7bff fbff 7bff 7bff 7c00 fc00 7e00(NaN) 3c00.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __attribute__((ext_vector_type(2))) float f2;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;
__global__ void probe(const float* x, unsigned* pk, unsigned* sc, int ovfl) {
    if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
    const int i = threadIdx.x;
    const f2 f = {x[i], x[i]};
    pk[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(f, h2)) & 0xffffu;      // v_cvt_pk_f16_f32
    _Float16 h;
    asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(h) : "v"(x[i]));
    sc[i] = (unsigned)__builtin_bit_cast(unsigned short, h);
}
int main() {
    const int n = 8;
    float hx[n] = {1e5f, -1e5f, 65504.f, 65520.f, INFINITY, -INFINITY, NAN, 1.0f};
    float* dx; unsigned *dp, *ds;
    hipMalloc(&dx, sizeof(hx)); hipMalloc(&dp, n * 4); hipMalloc(&ds, n * 4);
    hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
    for (int ovfl = 0; ovfl < 2; ++ovfl) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dx, dp, ds, ovfl);
        unsigned hp[n], hs[n];
        hipMemcpy(hp, dp, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hs, ds, n * 4, hipMemcpyDeviceToHost);
        printf("FP16_OVFL=%d  cvt_pk:", ovfl);
        for (int i = 0; i < n; ++i) printf(" %04x", hp[i]);
        printf("   cvt:");
        for (int i = 0; i < n; ++i) printf(" %04x", hs[i]);
        printf("\n");
    }
    return 0;
}

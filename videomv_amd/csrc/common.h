// common.h — shared device helpers for the gfx950 kernels (wave64, 16-bit storage (fp16 or bf16 per build) / fp32 math).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vmv.h"
#include <atomic>
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: set it once per (kernel, device), thread-safely
// (a function-local `static bool` skipped the second GPU of a process and raced between host threads).
// vmv_gemm_validate(): the whole host side of a launch — argument checks, tile policy, every launcher's own eligibility tests —
// with the device left alone.  A thread-local flag makes VMV_LAUNCH / vmv_lds_attr_once / vmv_launch_status no-ops, so the answer
// needs no GPU (the CPU test tier validates the packaged tile table with it).
extern thread_local int vmv_dry_run;
#define VMV_LAUNCH(kernel, grid, block, lds, st, ...) do { if (!vmv_dry_run) hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__); } while (0)
inline int vmv_lds_attr_once(std::atomic<unsigned long long>& done, const void* fn, int bytes) {
    if (vmv_dry_run) return 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return 0;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    done.fetch_or(bit, std::memory_order_release);
    return 0;
}


typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

#define VMV_DEV __device__ __forceinline__

// ---- the 16-bit storage / MFMA operand type of this build of the library (vmv.h: vmv_elem_type()).
// The same sources are compiled twice: -DVMV_BUILD_BF16 -> libvmv_hip_bf16.so (bf16, 8 significand bits),
// default -> libvmv_hip_f16.so (IEEE fp16, 11 significand bits; what the parity tolerances of DESIGN §6 are stated for).
// Both convert in hardware on gfx950 (v_cvt_pk_{bf16,f16}_f32, round-to-nearest-even) and run the same-rate MFMA.
#if defined(VMV_BUILD_BF16)
#define VMV_ELEM_TYPE 1
typedef __attribute__((ext_vector_type(8))) __bf16 elem8_t;
typedef __attribute__((ext_vector_type(2))) __bf16 elem2_t;
#define VMV_MFMA16 __builtin_amdgcn_mfma_f32_16x16x32_bf16
VMV_DEV float elem_lo(uint32_t w) { return __uint_as_float(w << 16); }
VMV_DEV float elem_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
VMV_DEV float elem_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
#else
#define VMV_ELEM_TYPE 0
typedef __attribute__((ext_vector_type(8))) _Float16 elem8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 elem2_t;
#define VMV_MFMA16 __builtin_amdgcn_mfma_f32_16x16x32_f16
VMV_DEV float elem_lo(uint32_t w) { return (float)__builtin_bit_cast(elem2_t, w).x; }
VMV_DEV float elem_hi(uint32_t w) { return (float)__builtin_bit_cast(elem2_t, w).y; }
VMV_DEV float elem_to_f32(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
#endif
// ---- fp16 stores SATURATE ("fp16 epilogues do not saturate; the only guard is a post-hoc finite check").  A finite fp32
// value beyond +-65504 becomes +-65504 in the 16-bit store instead of +-inf (true infinities and NaNs are preserved, so a genuinely
// broken forward still trips the finite check); bf16 has fp32's exponent range and needs nothing.  Two implementations, same result
// (tests/test_kernels_gpu.py::test_fp16_stores_saturate; tools/experiments/f16_ovfl_probe.hip shows the hardware honouring both):
//   VMV_F16_SAT = 1 (default): the wave's MODE.FP16_OVFL bit, set by the first instruction of every kernel (VMV_KERNEL_ENTER) —
//                 "an overflowed FP16 result is clamped to +-MAX_FP16 regardless of round mode, while still preserving true INF
//                 values" — which covers every v_cvt_pk_f16_f32 / v_cvt_f16_f32 of the kernel at zero cost per store;
//                 Measured side effect (tools/experiments/nan_probe.py, MI355X): with the bit set the fp16 MFMA no longer propagates
//                 non-finite OPERANDS — a NaN element contributes 0, an inf element the largest finite value (fp32 results clamp at
//                 FLT_MAX) — while the VALU paths (GroupNorm / LayerNorm statistics, softmax) still do.  Finite inputs and weights
//                 therefore cannot produce inf / NaN inside a forward, and a NaN that ENTERS one would be swallowed by the first
//                 GEMM: the host checks inputs and packed weights for finiteness at the door (diffusion_ddim._check_finite,
//                 packing.check_finite_weights) instead of relying on the result alone.
//   VMV_F16_SAT = 2: a v_med3_f32 clamp per converted value in pack_elem2 (two more VALU operations per stored pair; the GEGLU
//                 epilogues are VALU-issue-bound — only if a future part drops the mode bit);   VMV_F16_SAT = 0: round-3 behaviour.
#ifndef VMV_F16_SAT
#define VMV_F16_SAT 1
#endif
#if !defined(VMV_BUILD_BF16) && VMV_F16_SAT == 1
#define VMV_KERNEL_ENTER() asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1")
#else
#define VMV_KERNEL_ENTER() ((void)0)
#endif
VMV_DEV uint32_t pack_elem2(float lo, float hi) {
#if !defined(VMV_BUILD_BF16) && VMV_F16_SAT == 2
    // (infinities pass: med3(-65504, inf, 65504) would clamp them, so they are kept apart by the compare-free trick of clamping only
    //  finite values — v_med3 of a NaN returns the NaN-free median, hence the explicit class test)
    const float lo_c = __builtin_fminf(__builtin_fmaxf(lo, -65504.f), 65504.f), hi_c = __builtin_fminf(__builtin_fmaxf(hi, -65504.f), 65504.f);
    lo = (__builtin_isinf(lo) || lo != lo) ? lo : lo_c;
    hi = (__builtin_isinf(hi) || hi != hi) ? hi : hi_c;
#endif
    const f32x2_t f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, elem2_t));
}
// acc += a.lo * b.lo + a.hi * b.hi on packed element pairs (v_dot2c_f32_{f16,bf16}); volatile so that the caller's placement
// between MFMAs survives (the compiler otherwise sinks a fragment's sums into one block behind the MFMA batch)
#if defined(VMV_BUILD_BF16)
#define VMV_ELEM_ONE2 0x3f803f80u
VMV_DEV void elem_dot2c(float& acc, uint32_t a, uint32_t b) { asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b)); }
#else
#define VMV_ELEM_ONE2 0x3c003c00u
VMV_DEV void elem_dot2c(float& acc, uint32_t a, uint32_t b) { asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b)); }
#endif
VMV_DEV uint32_t f32_to_elem_bits(float f) { return pack_elem2(f, 0.f) & 0xffffu; }

VMV_DEV void unpack8(const u32x4_t& v, float* f) {
    f[0] = elem_lo(v.x); f[1] = elem_hi(v.x);
    f[2] = elem_lo(v.y); f[3] = elem_hi(v.y);
    f[4] = elem_lo(v.z); f[5] = elem_hi(v.z);
    f[6] = elem_lo(v.w); f[7] = elem_hi(v.w);
}
VMV_DEV u32x4_t pack8(const float* f) {
    u32x4_t v;
    v.x = pack_elem2(f[0], f[1]); v.y = pack_elem2(f[2], f[3]);
    v.z = pack_elem2(f[4], f[5]); v.w = pack_elem2(f[6], f[7]);
    return v;
}

// ---- max / cross-lane helpers of the softmax kernels.  Plain instructions through asm (not volatile: free to schedule):
// fmaxf() lowers to llvm.maxnum, which in the kernels' IEEE mode first canonicalises each operand with a v_max x, x.
VMV_DEV float vmax2(float a, float b) { float d; asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
VMV_DEV float vmax3(float a, float b, float c) { float d; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
// all-reduce steps over the lane pairs l ^ 16 and l ^ 32 with gfx950's VALU lane swaps (v_permlane16_swap exchanges lanes
// 16-31 / 48-63 of its first operand with lanes 0-15 / 32-47 of its second; v_permlane32_swap the upper half of the first
// with the lower half of the second): called with the same value twice, the two results hold, in every lane, the lane's and
// its partner's value — no LDS round trip as with ds_bpermute
typedef __attribute__((ext_vector_type(2))) unsigned int vmv_u2_t;
VMV_DEV float xor16_max(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const vmv_u2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return vmax2(__uint_as_float(r.x), __uint_as_float(r.y));
#else
    return x;
#endif
}
VMV_DEV float xor32_max3(float x, float c) {          // max(x, partner's x, c)
#if defined(__HIP_DEVICE_COMPILE__)
    const vmv_u2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return vmax3(__uint_as_float(r.x), __uint_as_float(r.y), c);
#else
    return x > c ? x : c;
#endif
}
VMV_DEV float xor16_32_sum(float x) {                 // sum over lanes l, l ^ 16, l ^ 32, l ^ 48
#if defined(__HIP_DEVICE_COMPILE__)
    // (not on any inner loop: written with its own wait states — gemm_rs.hip found hipcc's hazard pad in front of a
    //  v_permlane*_swap one state short when two swaps follow each other, stale data in lanes 28-31 / 60-63)
    float a = x, b = x;
    asm("s_nop 3\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    x = a + b;
    a = x; b = x;
    asm("s_nop 3\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
#else
    return x;
#endif
}

VMV_DEV float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// gelu(x) = x * Phi(x), the exact-erf form (F.gelu default), as
//     gelu(x) = max(x, 0) - |x| * Q(|x|),   Q(a) = 1 - Phi(a) = 2 ^ P(a),
// with P a degree-5 polynomial fitted (weighted minimax, a in [0, 6.5]) to log2 of the Gaussian upper tail: max abs error of the
// whole expression 6.7e-7 in fp32 arithmetic (tools/experiments/gelu_fit.py; beyond 6.5 the clamped term is < 3e-10), far
// below the 16-bit output ulp.  6 FMA-class operations + 1 transcendental (v_exp_f32 IS 2^x) + min / max: the GEGLU epilogues
// are VALU-issue-bound (round 3 stamps: the GELUs of a pair of output tiles took longer than the pair's 80 MFMAs), and this is
// half the issue slots of the Abramowitz-Stegun 7.1.26 erf (1 rcp + 1 exp + ~14 operations) it replaces.
VMV_DEV float gelu_erf_f(float x) {
    float a, m;
    asm("v_min_f32 %0, |%1|, %2" : "=v"(a) : "v"(x), "s"(6.5f));
    float L = -4.772448488e-04f;
    L = fmaf(L, a, 7.111470865e-03f);
    L = fmaf(L, a, -5.189299288e-02f);
    L = fmaf(L, a, -4.599231213e-01f);
    L = fmaf(L, a, -1.150818225e+00f);
    L = fmaf(L, a, -1.000033544e+00f);
    const float q = __builtin_amdgcn_exp2f(L);
    asm("v_max_f32 %0, 0, %1" : "=v"(m) : "v"(x));
    return fmaf(-a, q, m);
}

// VMV_ACT_* on four accumulator values (bias / rowvec already added): SiLU (the UNet's convs), exact-erf GELU (the CLIP text tower's MLP)
VMV_DEV void act_apply(f32x4_t& v, const int act) {
    if (act == VMV_ACT_SILU) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
    else if (act == VMV_ACT_GELU) { v.x = gelu_erf_f(v.x); v.y = gelu_erf_f(v.y); v.z = gelu_erf_f(v.z); v.w = gelu_erf_f(v.w); }
}

VMV_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

VMV_DEV bool vmv_ptr_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int vmv_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
static inline int vmv_launch_status() {
    if (vmv_dry_run) return VMV_OK;
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VMV_OK : (int)e;
}

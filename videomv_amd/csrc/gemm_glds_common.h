// gemm_glds_common.h — shared by the LDS-DMA GEMM kernels (gemm_glds.hip: one tile per block; gemm_pglds.hip: persistent).
#pragma once
#include "gemm_common.h"

namespace vmvg {

template <int WMW, int WN, int STAGES>
struct GlCfg {
    static constexpr int NW = 2 * WMW;                     // waves per block (WMW along M x 2 along N)
    static constexpr int NT = 64 * NW;
    static constexpr int BM = 64 * WMW;
    static constexpr int BN = 32 * WN;
    static constexpr int A_BYTES = BM * 128;
    static constexpr int W_BYTES = BN * 128;
    static constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
    static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;
    static constexpr int NAI = BM / (8 * NW);              // A wave-instructions per wave per chunk (8 rows each)
    static constexpr int NWI = (BN / 8 + NW - 1) / NW;     // W wave-instructions per wave per chunk
    static constexpr int LPT = NAI + NWI;                  // loads per lane per chunk
};

// 16-byte LDS-DMA through a buffer descriptor: lane address = base + voff + soff; a lane whose voff is out of range
// (>= num_records) WRITES ZEROS to its LDS slot (verified on gfx950: tools/experiments/buffer_lds_oob.hip) — this is
// how conv zero padding and the M / N / K tails are produced without a select on 64-bit pointers.
// (the builtin is only visible to the device pass: hipcc's host pass otherwise silently drops the kernel template's
//  instantiation — stub and handle come out undefined — so the body is compiled for the device only)
VMV_DEV void blds16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lptr, uint32_t voff, uint32_t soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lptr), 16, voff, soff, 0, 0);
#endif
}
#define VMV_BLDS16(rsrc, lptr, voff, soff) blds16(rsrc, lptr, voff, soff)
// 4-byte LDS-DMA (bias / column-sum / row-statistic strips): lane l lands at lptr + 4 l, OOB lanes write zeros
VMV_DEV void blds4(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lptr, uint32_t voff, uint32_t soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lptr), 4, voff, soff, 0, 0);
#endif
}
constexpr uint32_t OOB = 0x80000000u;          // > num_records of every descriptor below
constexpr uint32_t SRD_RECORDS = 0x7ffffff0u;
constexpr uint32_t SRD_FLAGS = 0x00020000u;

template <int N> VMV_DEV void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else static_assert(N == 0, "add the literal");
}

// run-time count (uniform): the literal must be an immediate, hence the switch
VMV_DEV void wait_vmcnt_rt(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
        case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
        case 23: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break;
        case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

}  // namespace vmvg

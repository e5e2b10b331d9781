// gemm_wreg.hip — EXPERIMENT, built only with `make EXPERIMENTS=1` (round 6).  Correct (it passes the
// linear / GEGLU / folded-LayerNorm GEMM tests as tile id 29) but SLOWER than the 8-wave wide tile: 582 / 637 TFLOP/s against 860 / 921 on the
// plain 7680 x 3840 / 10240 x 1280 products (profiles/r6_wreg_bench_v1.log) — an iteration takes ~4 000 cycles for 1 024 cycles of MFMAs: with the
// global loads staged through REGISTERS only two 32-KB chunks (64 VGPRs) fit in flight beside 256 accumulators (AGPRs) and 128 VGPRs of
// double-buffered fragments, i.e. ~2 000 cycles of cover for a loaded memory latency of twice that; a third chunk in flight needs 96 + 128 +
// addressing > 256 VGPRs.  (And round 2 measured that a denser GEMM lowers the clock of the kernels around it.)  Kept for the record.
//
// WIDE-WAVE, REGISTER-STAGED implicit GEMM for the long plain linears: the
// K = 1280 transformer linears of the third level (q | k | v 7680 x 3840, GEGLU-up 7680 x 10240) run at 0.71-0.73 x of hipBLASLt on the 8-wave
// tile kernels.  What the counters and the arithmetic say bounds those kernels is the LDS port: a 64 x 128 wave tile reads 12 fragments per
// 32 MFMAs, eight waves pull 96 KB per 32-deep k-step = 768 cycles of the 128-B/clk port against 1 024 cycles of MFMAs per SIMD.  Here a
// block is FOUR waves — one per SIMD, the whole 512-entry register file each — and a wave owns 128 x 128 of a 256 x 256 block tile:
// 16 fragments per 64 MFMAs (0.25 per MFMA instead of 0.375), 64 KB per k-step.  Round 2's 4-wave kernel (tools/experiments/gemm_wglds.hip)
// fed LDS by DMA and lost half its speed to the DMA issue slots nothing covered; this one loads global -> registers -> ds_write, so every
// memory instruction is a one-issue-slot affair that the wave itself threads between its MFMAs:
//   iteration t (chunk t = one 32-deep k-step, 64 MFMAs): the 16 fragment reads of chunk t + 1 (LDS stage (t + 1) % 3), the 8 ds_writes of
//   chunk t + 2 (from the registers its global loads were issued into two iterations ago) and the 8 global loads of chunk t + 4 are issued
//   one per two MFMAs; one block barrier per iteration.  The compiler's own wait counts order the loads (in-order vmcnt), the schedule is
//   pinned with sched_barrier.
// Plain linear segments only (one source, K % 32 == 0); every epilogue of gemm_common.h's epilogue_store (bias, folded LayerNorm by
// rowstat, GEGLU, rowvec, activation, residual, fp32 output).
#include "gemm_glds_common.h"
#include <cstdlib>
#include <type_traits>

using namespace vmvg;

namespace {

constexpr int WR_BM = 256, WR_BN = 256, WR_NT = 256, WR_S = 3;
constexpr int WR_TILE = 256 * 64;                 // one operand tile of a stage: 256 rows x 32 elements
constexpr int WR_STAGE = 2 * WR_TILE;             // A then W
constexpr int WR_LDS = WR_S * WR_STAGE;           // 96 KB
#if defined(VMV_BUILD_BF16)
#define WR_MFMA "v_mfma_f32_16x16x32_bf16"
#else
#define WR_MFMA "v_mfma_f32_16x16x32_f16"
#endif

__global__ __launch_bounds__(256, 1) void gemm_wreg_kernel(const VmvGemmParams p, const int tiles_m, const int tiles_n, const int nsteps, const int gm) {
    VMV_KERNEL_ENTER();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 15, fgrp = lane >> 4;

    // ---- XCD-aware, grouped tile order (gemm_xglds.hip)
    const int nblk = tiles_m * tiles_n;
    int logical;
    {
        const int bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tm_, tn_;
    if (gm > 1) {
        const int gsz = gm * tiles_n, g = logical / gsz, first = g * gm;
        const int gmh = tiles_m - first < gm ? tiles_m - first : gm;
        const int rem = logical - g * gsz;
        tn_ = rem / gmh; tm_ = first + (rem - tn_ * gmh);
    } else {
        tm_ = logical / tiles_n; tn_ = logical - tm_ * tiles_n;
    }
    const int m0 = tm_ * WR_BM, n0 = tn_ * WR_BN;

    // ---- loader: unit id = 256 j + tid (j = 0..3) -> row id >> 2, 16-byte k-slot id & 3; LDS image: row * 64 + (slot ^ ((row >> 2) & 3)) * 16
    const VmvGemmSeg& sg = p.seg[0];
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(sg.src), 0, (uint32_t)p.M * (uint32_t)sg.ld * 2u, SRD_FLAGS);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, (uint32_t)p.N * (uint32_t)p.ktot * 2u, SRD_FLAGS);
    uint32_t avo[4], wvo[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int id = 256 * j + tid, row = id >> 2, slot = id & 3;
        avo[j] = (m0 + row < p.M) ? (uint32_t)(((m0 + row) * sg.ld + slot * 8) * 2) : OOB;
        wvo[j] = (n0 + row < p.N) ? (uint32_t)(((n0 + row) * p.ktot + slot * 8) * 2) : OOB;
        lo[j] = (uint32_t)(row * 64 + ((slot ^ ((row >> 2) & 3)) << 4));
    }
    u32x4_t ga[2][4], gw[2][4];                     // two chunks of global loads in flight
    auto gload = [&](const int c, auto set_tag) __attribute__((always_inline)) {
        constexpr int set = decltype(set_tag)::value;
        const bool ok = c < nsteps;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ga[set][j] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, ok ? avo[j] : OOB, (uint32_t)(c * 64), 0);
            gw[set][j] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, ok ? wvo[j] : OOB, (uint32_t)(c * 64), 0);
        }
    };
    auto lwrite1 = [&](const int c, const int set, const int u) __attribute__((always_inline)) {        // unit u of 8: A units 0-3, W units 4-7
        unsigned char* st = smem + (c % WR_S) * WR_STAGE;
        if (u < 4) *reinterpret_cast<u32x4_t*>(st + lo[u]) = ga[set][u];
        else *reinterpret_cast<u32x4_t*>(st + WR_TILE + lo[u - 4]) = gw[set][u - 4];
    };
    // ---- fragments: rows 16 i + frow of the wave's 128-row halves, k-slot fgrp -> physical slot fgrp ^ ((frow >> 2) & 3)
    const uint32_t fo = (uint32_t)(frow * 64 + ((fgrp ^ ((frow >> 2) & 3)) << 4));
    u32x4_t fa[2][8], fw[2][8];
    auto fread1 = [&](const int c, const int buf, const int u) __attribute__((always_inline)) {         // fragment u of 16: A fragments 0-7, W fragments 8-15
        const unsigned char* st = smem + (c % WR_S) * WR_STAGE;
        if (u < 8) fa[buf][u] = *reinterpret_cast<const u32x4_t*>(st + (wm * 128 + 16 * u) * 64 + fo);
        else fw[buf][u - 8] = *reinterpret_cast<const u32x4_t*>(st + WR_TILE + (wn * 128 + 16 * (u - 8)) * 64 + fo);
    };

    f32x4_t acc[8][8];                              // [column tile j][row fragment i]
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: chunks 0 and 1 into their stages, chunks 2 and 3 in flight, fragments of chunk 0 in registers
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    gload(0, S0{});
    gload(1, S1{});
#pragma unroll
    for (int u = 0; u < 8; ++u) lwrite1(0, 0, u);
#pragma unroll
    for (int u = 0; u < 8; ++u) lwrite1(1, 1, u);
    gload(2, S0{});
    gload(3, S1{});
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int u = 0; u < 16; ++u) fread1(0, 0, u);

    // ---- main loop: two iterations per trip (fragment / staging parities are compile-time), NO branch in the body: chunks past the
    //      end of K are loaded through an out-of-range offset (zeros), so the guards are one v_cndmask per load, an odd chunk count
    //      costs one all-zero iteration, and the body exists exactly twice (a guarded tail instantiation made the allocator shuffle the
    //      256 accumulators between AGPRs and VGPRs: 686 spilled registers).
    auto iteration = [&](const int t, auto par_tag) __attribute__((always_inline)) {
        constexpr int cur = decltype(par_tag)::value, nxt = cur ^ 1;
        const bool more4 = t + 4 < nsteps;
#pragma unroll
        for (int q = 0; q < 32; ++q) {              // 32 slots: one memory instruction, two MFMAs
            if (q < 16) fread1(t + 1, nxt, q);
            else if (q < 24) lwrite1(t + 2, cur, q - 16);
            else {
                const int j = (q - 24) >> 1;
                if ((q & 1) == 0) ga[cur][j] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, more4 ? avo[j] : OOB, (uint32_t)((t + 4) * 64), 0);
                else gw[cur][j] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, more4 ? wvo[j] : OOB, (uint32_t)((t + 4) * 64), 0);
            }
#pragma unroll
            for (int e = 2 * q; e < 2 * q + 2; ++e) {
                const int i = e >> 3, j = e & 7;
                // (accumulators pinned to AGPRs, in place: left to the allocator the 256 of them wander between the two register files —
                //  295 v_accvgpr moves and 35 scratch accesses inside the loop.  64 independent accumulators: no MFMA -> MFMA hazard.)
                asm volatile(WR_MFMA " %0, %1, %2, %0" : "+a"(acc[j][i]) : "v"(fw[cur][j]), "v"(fa[cur][i]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);        // my ds_writes of chunk t + 2 (and fragment reads of chunk t + 1) are done
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    for (int t = 0; t < nsteps; t += 2) {
        iteration(t, std::integral_constant<int, 0>{});
        iteration(t + 1, std::integral_constant<int, 1>{});
    }

    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");       // (the last MFMAs' results are read below: inline asm gets no hazard padding)
    // ---- epilogue: lane owns row m = .. + frow, channels n = .. + 4 fgrp + {0..3} of every (i, j) tile
    const int mbase = m0 + wm * 128 + frow;
    const int nbase = n0 + wn * 128 + 4 * fgrp;
    if (p.epilogue == VMV_EPI_GEGLU) {
#pragma unroll
        for (int j = 0; j < 8; j += 2)
#pragma unroll
            for (int i = 0; i < 8; ++i) { epilogue_store(p, mbase + 16 * i, nbase + 16 * j, acc[j][i], acc[j + 1][i]); __builtin_amdgcn_sched_barrier(0); }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) { epilogue_store(p, mbase + 16 * i, nbase + 16 * j, acc[j][i], acc[j][i]); __builtin_amdgcn_sched_barrier(0); }
    }
}

}  // namespace

bool vmv_gemm_wreg_supported(const VmvGemmParams& p) {
    if (p.nseg != 1 || p.seg[0].mode != VMV_SEG_LINEAR || p.seg[0].k != p.ktot) return false;
    if ((p.ktot & 31) || p.ktot < 128) return false;
    if (p.ksplit > 1 || p.wgroup_rows != 0 || p.gn_table || p.gn_silu || vmv_gemm_ln_inline(p) || p.epilogue == VMV_EPI_TATTN) return false;
    if ((long)(p.M + 256) * p.seg[0].ld * 2 >= (1L << 31) - 65536 || (long)(p.N + 256) * p.ktot * 2 >= (1L << 31) - 65536) return false;
    return true;
}

int vmv_gemm_wreg_launch(const VmvGemmParams& p, hipStream_t st) {
    if (!vmv_gemm_wreg_supported(p)) return VMV_GLDS_UNSUPPORTED;
    const int tiles_m = (p.M + WR_BM - 1) / WR_BM, tiles_n = (p.N + WR_BN - 1) / WR_BN;
    static int gm_env = -2;
    if (gm_env == -2) { const char* e = getenv("VMV_WREG_GM"); gm_env = e ? atoi(e) : -1; }
    int gm = gm_env >= 1 ? gm_env : 1;
    if (gm_env < 1 && tiles_m >= 2 && tiles_n >= 2) {
        int best_cost = 1 + 32;
        for (int g = 2; g <= 32; g *= 2) {
            const int gn = (32 + g - 1) / g;
            if (gn > tiles_n || g > tiles_m) continue;
            if (g + gn < best_cost) { gm = g; best_cost = g + gn; }
        }
    }
    static std::atomic<unsigned long long> attr{0};
    if (const int rc = vmv_lds_attr_once(attr, reinterpret_cast<const void*>(&gemm_wreg_kernel), WR_LDS)) return rc;
    VMV_LAUNCH(gemm_wreg_kernel, dim3(tiles_m * tiles_n), dim3(WR_NT), WR_LDS, st, p, tiles_m, tiles_n, p.ktot / 32, gm);
    return vmv_launch_status();
}

// gemm_rs.hip — ROW-STATIONARY implicit GEMM for the short-K linears of the transformer blocks (same contract as gemm.hip /
// vmv.h; K = 320 / 640 = the channel counts of the UNet's two large levels, and K = 512 = the init TemporalTransformer's 8 x 64 heads).
//
// Why: at K = C the tile kernels spend a tile's life outside their main loop.  A 256 x 128 tile of the K = 320 GEGLU is five
// 64-deep chunks: per chunk the CU's LDS-DMA path moves 48 KB (A + W) for 1 k cycles of MFMAs, the epilogue is as long as
// the loop, and the result (profiles/r2_ops_40x64.tsv) is 19-28 % of the MFMA peak at L0 with every unit half idle.  These
// GEMMs are also close to the HBM roofline (qkv: 315 MB for 75 GFLOP), so what counts is bytes through the CU per MAC.
// Here the ACTIVATIONS never enter LDS:
//   * a wave keeps RT x 16 rows of A — the whole K range — in registers (RT = 4, K = 320: 160 registers; loaded once,
//     fragment-shaped, straight from HBM / L2);
//   * a block of 8 waves (RT x 128 rows) streams the weight matrix through a three-stage LDS ring in 40-KB chunks
//     (64 W rows at K = 320) by LDS-DMA: per MAC 1/512 of a byte instead of 1/256 + 1/128, and the LDS port only serves W
//     fragments (one ds_read_b128 per RT MFMAs);
//   * K is complete inside the wave, so an output tile is FINISHED after K/32 MFMAs per row tile: there is no accumulator
//     array to drain, no tile epilogue — bias, GEGLU, residual and the store follow each pair of 16-column tiles while the
//     SIMD's other wave multiplies; two lane swaps (v_permlane16_swap) turn the MFMA's 4-channel lane slices into 16-byte
//     stores (64 B per row per instruction);
//   * a LayerNorm folded into the GEMM (vmv.h: colsum / ln_eps) is applied to the resident rows before the first MFMA:
//     two-pass statistics from the registers, x <- (x - mean) * rstd, then the plain product with W' = W diag(gamma) —
//     no statistics launch, no rowstat traffic, no colsum correction in the epilogue.
// One barrier per chunk (160 MFMAs per wave); the ring runs two chunks ahead.
#include "gemm_glds_common.h"
#include <cstdlib>
#include <type_traits>

using namespace vmvg;

namespace {

template <int RT, int KS>
struct RsCfg {
    static constexpr int NW = 8, NT = 512;
    static constexpr int BM = NW * RT * 16;                    // rows per block
    static constexpr int K = KS * 32;
    static constexpr int RB = K * 2;                           // bytes per W row
    static constexpr int SPR = RB / 16;                        // 16-byte slots per W row (40 / 80)
    static constexpr int CHUNK_BYTES = KS == 16 ? 32768 : 40960;   // whole W rows: 64 x 640 B, 32 x 1280 B, 32 x 1024 B (K = 512: round 6)
    static constexpr int CROWS = CHUNK_BYTES / RB;             // W rows per chunk (64 / 32)
    static constexpr int G = CROWS / 16;                       // 16-row W tiles per chunk (4 / 2)
    static constexpr int STAGES = 3;
    static constexpr int PIECES = CHUNK_BYTES / 1024 / NW;     // LDS-DMA wave-instructions per wave per chunk (5; 4 at K = 512)
    static constexpr int MAX_COLS = 5120;                      // W rows one block walks (bias strip: 20 KB)
    static constexpr int TAB_BYTES = 2 * 2 * K * 4;            // folded GroupNorm: (scale | shift)[K] fp32 of the two stat groups a block can touch
    static constexpr int LDS_BYTES = STAGES * CHUNK_BYTES + MAX_COLS * 4 + TAB_BYTES;
    static_assert(CROWS * RB == CHUNK_BYTES && (G == 2 || G == 4) && PIECES * NW * 1024 == CHUNK_BYTES, "chunk geometry");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// v_permlane16_swap on the dword pairs (x, z) and (y, w): lanes 16-31 / 48-63 of the first dword trade places with lanes
// 0-15 / 32-47 of the second (an involution: applied twice it is the identity).
// Inline asm with its own wait states: the builtin form compiled to `v_cvt_pk v191 / s_nop 0 / swap v188, v190 / swap v189, v191`,
// and on gfx950 the second swap then read the OLD v191 in the last four lanes of each half-wave (lanes 28-31, 60-63: output
// channels 2-3 of rows 12-15 of a tile came out as garbage, every launch) — hipcc's hazard pad for "VALU write -> v_permlane*_swap
// read" is one state short when another swap sits in between.  Four states in front cover both swaps; the operands of the
// asm statement are opaque to the scheduler, so nothing can slide in between.
VMV_DEV u32x4_t swap16_xz_yw(u32x4_t v) {
    uint32_t x = v.x, y = v.y, z = v.z, w = v.w;
    asm("s_nop 3\n\tv_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3\n\ts_nop 1" : "+v"(x), "+v"(y), "+v"(z), "+v"(w));
    return u32x4_t{x, y, z, w};
}

constexpr int RS_GEGLU = 1, RS_LN = 2, RS_RES = 4, RS_GN = 8;
#ifndef VMV_RS_STAGGER
#define VMV_RS_STAGGER 0        // experiments: 1 = waves 4-7 take the chunk barrier between the MFMAs and the epilogue of a chunk's last pair
#endif
#ifndef VMV_RS_RELAX
#define VMV_RS_RELAX 1          // experiments: 0 = the chunk-end wait lets only the current chunk's stores stay in flight
#endif
#ifndef VMV_RS_PFD
#define VMV_RS_PFD 1            // W-fragment prefetch distance in k-steps (experiments: 1, 2, 3)
#endif
#ifndef VMV_RS_ABLATE
#define VMV_RS_ABLATE 0         // experiments (results are wrong): 1 no stores, 2 GELU -> identity, 3 no MFMAs, 4 no W DMA after the prologue,
#endif                          // 5 no LayerNorm arithmetic, 6 no W fragment reads, 7 no chunk barriers; 8 = s_memtime stamps (results right)

template <int RT, int KS, int MODE>
__global__ __launch_bounds__(512, 1) void gemm_rs_kernel(const VmvGemmParams p, const int tiles_m, const int nsplit, const int cols_per_split) {
    VMV_KERNEL_ENTER();
    using Cfg = RsCfg<RT, KS>;
    constexpr bool GEGLU = (MODE & RS_GEGLU) != 0, LN = (MODE & RS_LN) != 0, RES = (MODE & RS_RES) != 0, GN = (MODE & RS_GN) != 0;
    constexpr int RB = Cfg::RB, G = Cfg::G, P = Cfg::PIECES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fgrp = lane >> 4;

    // ---- block -> (row tile, column split); XCD-aware bijection: the splits of one row tile run on one XCD (A comes from its L2)
    const int nblk = tiles_m * nsplit;
    int logical;
    {
        const int bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mt = logical / nsplit, ns = logical - mt * nsplit;
    const int m_wave = mt * Cfg::BM + wave * (16 * RT);
    const int n_begin = ns * cols_per_split;
    const int ncols = (p.N - n_begin) < cols_per_split ? (p.N - n_begin) : cols_per_split;     // W rows of this block: k * CROWS (launcher)
    const int NC = ncols / Cfg::CROWS;
#if VMV_RS_ABLATE == 8      // experiments: block 0, waves 0 and 4 stamp s_memtime at the phase boundaries into p.workspace
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(p.workspace);
    int stamp_i = 0;
    auto stamp = [&]() {
        if (stamps && blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4) && stamp_i < 256)
            stamps[(wave == 4 ? 256 : 0) + stamp_i] = __builtin_readcyclecounter();
        ++stamp_i;
    };
#define RS_STAMP() stamp()
#else
#define RS_STAMP()
#endif
    RS_STAMP();      // 0: kernel start

    // ---- the wave's RT x 16 rows of A, whole K range, straight into registers (rows >= M read as zero through the descriptor)
    const VmvGemmSeg& sg = p.seg[0];
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(sg.src), 0, (uint32_t)p.M * (uint32_t)sg.ld * 2u, SRD_FLAGS);
    u32x4_t a[RT][KS];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const uint32_t vo = (uint32_t)((m_wave + 16 * i + frow) * sg.ld + 8 * fgrp) * 2u;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) a[i][kk] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, vo + (uint32_t)(kk * 64), 0, 0);
    }

    // ---- loader: bias strip, then the W ring.  A chunk is CROWS whole rows of W (40 KB); wave w fills bytes [5120 w, 5120 (w + 1))
    //      of it with 5 wave-instructions.  LDS position (row r, 16-byte slot s') holds k-slot s' ^ swz(r) of that row, applied
    //      to the SOURCE address (the image of a DMA is lane-linear): with 640-byte rows the 16 rows of a fragment alternate
    //      between the two halves of the 256-byte bank row, so swz = (r >> 1) & 7 spreads them over all 16 16-byte units; with
    //      1280-byte rows every row starts on the same bank, swz = r & 15.  Both are conflict-free for ds_read_b128's
    //      non-contiguous 16-lane groups (rows {0-3, 12-15} with k-slot f, rows {4-11} with f ^ 1).
    float* bias_lds = reinterpret_cast<float*>(smem + Cfg::STAGES * Cfg::CHUNK_BYTES);
    {
        const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias ? (uint32_t)p.N * 4u : 0u, SRD_FLAGS);
        for (int q = wave; q * 256 < ncols; q += Cfg::NW)      // 256 floats per wave-instruction; columns >= N / no bias: zeros
            VMV_BLDS16(b_rsrc, reinterpret_cast<unsigned char*>(bias_lds) + q * 1024, (uint32_t)(n_begin + q * 256 + 4 * lane) * 4u, 0);
    }
    // folded GroupNorm (vmv.h: gn_table): a block's rows lie in at most two consecutive stat groups (gn_rows_per_stat >= BM); their
    // scale / shift rows — 2 x 2 x K floats, contiguous in the table — go to LDS behind the bias strip, before the W ring
    float* tab_lds = bias_lds + Cfg::MAX_COLS;
    int gn_first = 0;
    if constexpr (GN) {
        gn_first = (mt * Cfg::BM) / p.gn_rows_per_stat;
        const uint32_t nstat = (uint32_t)((p.M + p.gn_rows_per_stat - 1) / p.gn_rows_per_stat);
        const __amdgpu_buffer_rsrc_t t_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gn_table), 0, nstat * (uint32_t)(2 * Cfg::K * 4), SRD_FLAGS);
        for (int q = wave; q * 1024 < Cfg::TAB_BYTES; q += Cfg::NW)
            VMV_BLDS16(t_rsrc, reinterpret_cast<unsigned char*>(tab_lds) + q * 1024, (uint32_t)(gn_first * 2 * Cfg::K * 4 + q * 1024 + 16 * lane), 0);
    }
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, (uint32_t)p.N * (uint32_t)p.ktot * 2u, SRD_FLAGS);
    auto issue_chunk = [&](int c, int slot) {
        unsigned char* base = smem + slot * Cfg::CHUNK_BYTES + wave * (P * 1024);
        const uint32_t so = (uint32_t)((n_begin + c * Cfg::CROWS) * p.ktot) * 2u;
        // (the five source offsets are recomputed per chunk — ~30 VALU operations per 160+ MFMAs — instead of living in five
        //  registers next to 160 of resident rows; the empty asm keeps the compiler from hoisting them out of the loop again)
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const int u = wave * (P * 64) + q * 64 + ln;
            const int r = u / Cfg::SPR, s = u - r * Cfg::SPR;
            const int sw = KS == 10 ? ((r >> 1) & 7) : (r & 15);
            VMV_BLDS16(w_rsrc, base + q * 1024, (uint32_t)(r * p.ktot + (s ^ sw) * 8) * 2u, so);
        }
    };
    const int pro = NC < Cfg::STAGES ? NC : Cfg::STAGES;
    for (int c = 0; c < pro; ++c) issue_chunk(c, c);

    // ---- folded GroupNorm: x <- elem(x * scale[c] + shift[c]) — the values vmv_groupnorm_apply would have stored — on the resident rows
    if constexpr (GN) {
        wait_vmcnt_rt(pro * P);                      // everything issued before the W ring has landed (in-order): my A rows, my table pieces
        __syncthreads();                             // ... and every wave's table pieces
        // per row tile the LDS address of its stat group's table (scalar select), per (row tile, k-step) 16 table values; the loop is
        // fenced per step and reads one step ahead: left to itself the scheduler hoists every step's reads (600 spilled registers)
        const float* tbase[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int l = __builtin_amdgcn_readfirstlane((m_wave + 16 * i) / p.gn_rows_per_stat - gn_first);      // 0 or 1
            tbase[i] = tab_lds + (l ? 2 * Cfg::K : 0) + 8 * fgrp;
        }
        constexpr int GPF = 1, GNB = GPF + 1;        // table reads run GPF steps ahead (16 registers per step in flight; 2 measured the same)
        f32x4_t tv[GNB][4];
        int chain = 0;                               // always 0; ties each read to an earlier step's result (see the asm below)
        auto rd = [&](int idx, f32x4_t (&t)[4]) {
            const float* tb = tbase[idx / KS] + (idx % KS) * 32 + chain;
            t[0] = *reinterpret_cast<const f32x4_t*>(tb); t[1] = *reinterpret_cast<const f32x4_t*>(tb + 4);
            t[2] = *reinterpret_cast<const f32x4_t*>(tb + Cfg::K); t[3] = *reinterpret_cast<const f32x4_t*>(tb + Cfg::K + 4);
        };
#pragma unroll
        for (int idx = 0; idx < GPF; ++idx) rd(idx, tv[idx % GNB]);
#pragma unroll
        for (int idx = 0; idx < RT * KS; ++idx) {
            if (idx + GPF < RT * KS) rd(idx + GPF, tv[(idx + GPF) % GNB]);
            __builtin_amdgcn_sched_barrier(0);
            const f32x4_t sc0 = tv[idx % GNB][0], sc1 = tv[idx % GNB][1], sh0 = tv[idx % GNB][2], sh1 = tv[idx % GNB][3];
            u32x4_t v = a[idx / KS][idx % KS];
            v.x = pack_elem2(fmaf(elem_lo(v.x), sc0.x, sh0.x), fmaf(elem_hi(v.x), sc0.y, sh0.y));
            v.y = pack_elem2(fmaf(elem_lo(v.y), sc0.z, sh0.z), fmaf(elem_hi(v.y), sc0.w, sh0.w));
            v.z = pack_elem2(fmaf(elem_lo(v.z), sc1.x, sh1.x), fmaf(elem_hi(v.z), sc1.y, sh1.y));
            v.w = pack_elem2(fmaf(elem_lo(v.w), sc1.z, sh1.z), fmaf(elem_hi(v.w), sc1.w, sh1.w));
            a[idx / KS][idx % KS] = v;
            // (a fence for the optimiser, not only the scheduler: without a data dependency every step's table reads are issued up
            //  front — 640 live registers; the empty asm makes a later read's address "depend" on this step's result)
            asm volatile("" : "+v"(chain) : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- LayerNorm of the resident rows (two-pass, fp32): lanes frow, frow + 16, + 32, + 48 hold the four k-quarters of a row
    if constexpr (LN && VMV_RS_ABLATE != 5) {
        const float inv_k = 1.0f / (float)Cfg::K;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            float s1 = 0.f;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                elem_dot2c(s1, a[i][kk].x, VMV_ELEM_ONE2); elem_dot2c(s1, a[i][kk].y, VMV_ELEM_ONE2);
                elem_dot2c(s1, a[i][kk].z, VMV_ELEM_ONE2); elem_dot2c(s1, a[i][kk].w, VMV_ELEM_ONE2);
            }
            // (v_dot2c is a DOT-pipe instruction: on gfx940+ its result needs 3-4 wait states before a different VALU reads it,
            //  and hipcc pads nothing for an instruction it only sees as inline asm — without this the next v_mov picked up a
            //  stale partial sum: means off by ~1e-2 sigma on the first GPU run)
            asm volatile("s_nop 4" : "+v"(s1));
            const float mean = xor16_32_sum(s1) * inv_k;
            float s2 = 0.f;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const uint32_t w4[4] = {a[i][kk].x, a[i][kk].y, a[i][kk].z, a[i][kk].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d0 = elem_lo(w4[e]) - mean, d1 = elem_hi(w4[e]) - mean;
                    s2 = fmaf(d0, d0, s2); s2 = fmaf(d1, d1, s2);
                }
            }
            const float rstd = __builtin_amdgcn_rsqf(xor16_32_sum(s2) * inv_k + p.ln_eps);
            const float nm = -mean * rstd;
            // (the normalisation pass unpacks the same registers as the variance pass: left visible, the compiler keeps the 80
            //  unpacked fp32 values of the row tile alive in between — the bf16 build, whose unpack is a shift, spilled 300
            //  registers that way — so the packed registers are made opaque here)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) asm volatile("" : "+v"(a[i][kk].x), "+v"(a[i][kk].y), "+v"(a[i][kk].z), "+v"(a[i][kk].w));
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                a[i][kk].x = pack_elem2(fmaf(elem_lo(a[i][kk].x), rstd, nm), fmaf(elem_hi(a[i][kk].x), rstd, nm));
                a[i][kk].y = pack_elem2(fmaf(elem_lo(a[i][kk].y), rstd, nm), fmaf(elem_hi(a[i][kk].y), rstd, nm));
                a[i][kk].z = pack_elem2(fmaf(elem_lo(a[i][kk].z), rstd, nm), fmaf(elem_hi(a[i][kk].z), rstd, nm));
                a[i][kk].w = pack_elem2(fmaf(elem_lo(a[i][kk].w), rstd, nm), fmaf(elem_hi(a[i][kk].w), rstd, nm));
            }
        }
    }

    // ---- output / residual addressing: after the lane swaps a lane holds 8 consecutive channels of row frow:
    //      columns (fgrp & 1) * 16 + (fgrp >> 1) * 8 .. + 8 of the 32-column pair
    const int lanecol = (fgrp & 1) * 16 + (fgrp >> 1) * 8;
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, SRD_RECORDS, SRD_FLAGS);
    const __amdgpu_buffer_rsrc_t res_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.residual), 0, SRD_RECORDS, SRD_FLAGS);
    const uint32_t ovo = (uint32_t)(frow * p.ldo + lanecol) * 2u;
    // GEGLU (one 16-column output tile per pair): the swap pairs two ROW tiles instead — lanes with even fgrp end up with 8
    // channels of row frow of tile i, odd fgrp with those of tile i + 1; columns (fgrp >> 1) * 8 .. + 8
    const uint32_t ovo_g = (uint32_t)((frow + 16 * (fgrp & 1)) * p.ldo + (fgrp >> 1) * 8) * 2u;
    const uint32_t rvo = (uint32_t)(frow * p.ldr + lanecol) * 2u;
    const float rs = p.res_scale != 0.f ? p.res_scale : 1.f;
    auto row_ok = [&](int i) { return m_wave + 16 * i + frow < p.M; };

    RS_STAMP();      // 1: resident rows loaded (and normalised), DMA issued
    // ---- first chunk (and the bias strip, issued before it) visible to every wave
    if (pro >= 3) wait_vmcnt_rt(2 * P); else if (pro == 2) wait_vmcnt_rt(P); else wait_vmcnt_rt(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // fragment addressing: this lane's rows 16 j + frow have swz = fsw (lane constant); k-slot (4 kk + fgrp) ^ fsw =
    // 4 (kk ^ (fsw >> 2)) + (fgrp ^ (fsw & 3)), so with NB = 2 (4) lane offsets, one per kk mod NB, every read is base + 64 kk
    RS_STAMP();      // 2: first chunk landed, barrier passed
    const int fsw = KS == 10 ? ((frow >> 1) & 7) : frow;      // swz() of this lane's fragment rows
    constexpr int NB = KS == 10 ? 2 : 4;
    int foff[NB];
#pragma unroll
    for (int r = 0; r < NB; ++r) foff[r] = frow * RB + (fgrp ^ (fsw & 3)) * 16 + ((r ^ (fsw >> 2)) - r) * 64;
    auto mma_pair = [&](const unsigned char* sbase, const int q, f32x4_t (&c0)[RT], f32x4_t (&c1)[RT]) {
        const unsigned char* tb[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r) tb[r] = sbase + 32 * q * RB + foff[r];
#pragma unroll
        for (int i = 0; i < RT; ++i) { c0[i] = f32x4_t{0.f, 0.f, 0.f, 0.f}; c1[i] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
        // W fragments of k-step kk + PFD are requested before the MFMAs of k-step kk (the order is pinned: left alone the compiler
        // issues each pair of reads right in front of its MFMAs and the matrix pipe waits out the LDS latency 10 times per pair;
        // ablation on the GPU: with the reads removed the kernel runs 36-47 % faster, i.e. one k-step of cover — 8 MFMAs = 128
        // cycles at 64 rows per wave, 64 cycles at 32 — is less than the loaded LDS latency)
        // (the residual variant at 64 rows per wave has no registers left for a second fragment set: it is HBM-bound anyway)
        constexpr int PFD = (RES && RT == 4) ? 0 : (VMV_RS_PFD);
        constexpr int NBUF = PFD + 1;
        u32x4_t w0[NBUF], w1[NBUF];
        auto rd = [&](const int kk, u32x4_t& x0, u32x4_t& x1) {
            const unsigned char* t = tb[kk & (NB - 1)] + 64 * kk;
#if VMV_RS_ABLATE == 6
            x0 = u32x4_t{(uint32_t)(uintptr_t)t, 1u, 2u, 3u}; x1 = x0;
#else
            x0 = *reinterpret_cast<const u32x4_t*>(t);
            x1 = *reinterpret_cast<const u32x4_t*>(t + 16 * RB);
#endif
        };
#pragma unroll
        for (int kk = 0; kk < PFD; ++kk) rd(kk, w0[kk % NBUF], w1[kk % NBUF]);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int cur = kk % NBUF;
            if (kk + PFD < KS) rd(kk + PFD, w0[(kk + PFD) % NBUF], w1[(kk + PFD) % NBUF]);
            if constexpr (PFD > 0) __builtin_amdgcn_sched_barrier(0);
#if VMV_RS_ABLATE == 3
            c0[0].x += __uint_as_float(w0[cur].x ^ a[kk % RT][kk].x); c1[0].x += __uint_as_float(w1[cur].y ^ a[(kk + 1) % RT][kk].y);
#else
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                c0[i] = VMV_MFMA16(__builtin_bit_cast(elem8_t, w0[cur]), __builtin_bit_cast(elem8_t, a[i][kk]), c0[i], 0, 0, 0);
                c1[i] = VMV_MFMA16(__builtin_bit_cast(elem8_t, w1[cur]), __builtin_bit_cast(elem8_t, a[i][kk]), c1[i], 0, 0, 0);
            }
#endif
            if constexpr (PFD > 0) __builtin_amdgcn_sched_barrier(0);
        }
    };
    // residual of one pair in store layout, requested before the pair's MFMAs
    auto load_res = [&](const int ocol, u32x4_t (&rv)[RT]) {
#pragma unroll
        for (int i = 0; i < RT; ++i)
            rv[i] = __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, row_ok(i) ? rvo : OOB, (uint32_t)((m_wave + 16 * i) * p.ldr + ocol) * 2u, 0);
    };
    auto store_pair = [&](const int ocol, const int i, u32x4_t o) {
        o = swap16_xz_yw(o);
#if VMV_RS_ABLATE == 1
        if (o.x == 0x12345u && o.y == 0x54321u)
#endif
        __builtin_amdgcn_raw_buffer_store_b128(o, out_rsrc, row_ok(i) ? ovo : OOB, (uint32_t)((m_wave + 16 * i) * p.ldo + ocol) * 2u, 0);
        // Store-data discipline (cf. gemm_pglds.hip): the allocator hands the store's registers to the next row tile's
        // v_pk_add_f32 at once, and with the VALU write directly behind the store the LAST dword of the last four lanes of
        // every 16-lane row went out holding the new fp32 value (first GPU run of this kernel: every launch, rows 12-15 of a
        // tile, channels 6-7 of each 8).  The data stays live — and nothing is issued — for 8 states after the store.
        asm volatile("s_nop 7" ::"v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w) : "memory");
    };

    // body b: W rows [64 b, 64 b + 64) of the block = two pairs; one chunk (G = 4) or two (G = 2)
    const int nbody = ncols / 64;
    int c = 0, slot = 0;
    auto chunk_end = [&](const int stores) {       // the wave's stores issued behind chunk c + 2's DMA (or a lower bound)
        if (c + 1 < NC) {
            // chunk c + 1 landed (mine): everything issued after its DMA — chunk c + 2's DMA and this chunk's stores — may stay in flight
#if VMV_RS_ABLATE == 4 || VMV_RS_ABLATE == 1
            wait_vmcnt_rt(0);
#else
            wait_vmcnt_rt((c + 2 < NC ? P : 0) + stores);
#endif
            __builtin_amdgcn_s_waitcnt(0xc07f);
#if VMV_RS_ABLATE != 7
            __builtin_amdgcn_s_barrier();          // every wave is done with slot `slot`, chunk c + 1 is visible
#endif
            asm volatile("" ::: "memory");
#if VMV_RS_ABLATE != 4
            if (c + Cfg::STAGES < NC) issue_chunk(c + Cfg::STAGES, slot);
#endif
        }
        ++c;
        slot = slot + 1 == Cfg::STAGES ? 0 : slot + 1;
    };
    // Staggered halves.  All eight waves meet at the chunk barrier, so left alone the two waves of a SIMD (w and w + 4) run in
    // lock-step: both multiply, then both run their epilogues with the matrix pipe idle (first GPU run: 8.8 k cycles per body of
    // 2 x 2.56 k MFMA cycles).  Waves 4-7 therefore take the barrier BETWEEN the MFMAs and the epilogue of a chunk's last pair
    // (their accumulators simply stay live across it): after every barrier one wave of the SIMD starts multiplying while the
    // other starts an epilogue, and they keep alternating.  The wait count of the late half names the stores that sit behind
    // the next chunk's DMA at that point (a smaller count than the true one only waits for more).
    constexpr bool STAGGER = VMV_RS_STAGGER != 0 && !RES;
    const bool late = STAGGER && wave >= Cfg::NW / 2;
    for (int b = 0; b < nbody; ++b) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned char* sbase = smem + slot * Cfg::CHUNK_BYTES;
            const int q = G == 4 ? h : 0;                         // pair index inside the chunk
            const int nrel = 64 * b + 32 * h;                     // first W row of the pair, relative to n_begin
            const int ocol = GEGLU ? (n_begin + nrel) / 2 : n_begin + nrel;        // first output column of the pair
            f32x4_t c0[RT], c1[RT];
            u32x4_t rv[RT];
            if constexpr (RES) load_res(ocol, rv);
            RS_STAMP();      // 3 + 6 k (+ 3 for h = 1): pair start
            mma_pair(sbase, q, c0, c1);
            RS_STAMP();      // MFMAs issued
            const bool cend = G == 2 || h == 1;                   // last pair of its chunk (compile-time after unrolling)
            constexpr int SP = GEGLU ? RT / 2 : RT;               // store instructions per pair
            // chunk_end(n): n = this wave's stores issued since chunk c + 1's DMA went out (at the chunk_end two chunks ago); they and
            // chunk c + 2's DMA may stay in flight.  (The first version waited for all but the last chunk's stores: with the
            // write path of a store-heavy GEMM backed up that is a write round trip on the critical path of every chunk.)
            if (cend && late) {
#if VMV_RS_RELAX
                if (G == 4) chunk_end(c == 0 ? SP : c == 1 ? 3 * SP : 4 * SP);
                else chunk_end(c == 0 ? 0 : c == 1 ? SP : 2 * SP);
#else
                if (G == 4) chunk_end(c == 0 ? SP : 2 * SP);
                else chunk_end(c == 0 ? 0 : SP);
#endif
            }
            const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(bias_lds + nrel + 4 * fgrp);
            const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(bias_lds + nrel + 16 + 4 * fgrp);
            if constexpr (GEGLU) {
                u32x2_t hp[RT];
#pragma unroll
                for (int i = 0; i < RT; ++i) {
                    f32x4_t v = c0[i] + b0;
                    const f32x4_t gt = c1[i] + b1;
#if VMV_RS_ABLATE == 2
                    v *= gt;
#else
                    v.x *= gelu_erf_f(gt.x); v.y *= gelu_erf_f(gt.y); v.z *= gelu_erf_f(gt.z); v.w *= gelu_erf_f(gt.w);
#endif
                    hp[i].x = pack_elem2(v.x, v.y); hp[i].y = pack_elem2(v.z, v.w);
                    if (i & 1) {
                        u32x4_t o = swap16_xz_yw(u32x4_t{hp[i - 1].x, hp[i - 1].y, hp[i].x, hp[i].y});
                        const bool ok = m_wave + 16 * (i - 1 + (fgrp & 1)) + frow < p.M;
#if VMV_RS_ABLATE == 1
                        if (o.x == 0x12345u && o.y == 0x54321u)
#endif
                        __builtin_amdgcn_raw_buffer_store_b128(o, out_rsrc, ok ? ovo_g : OOB, (uint32_t)((m_wave + 16 * (i - 1)) * p.ldo + ocol) * 2u, 0);
                        asm volatile("s_nop 7" ::"v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w) : "memory");       // (store-data discipline: store_pair)
                    }
                    // one row tile at a time: left alone the scheduler interleaves all 16 GELUs of the pair for ILP
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < RT; ++i) {
                    f32x4_t v0 = c0[i] + b0, v1 = c1[i] + b1;
                    if constexpr (RES) {
                        const u32x4_t r = swap16_xz_yw(rv[i]);   // store layout -> this lane's 4 channels of tile 2 q (x, y) and 2 q + 1 (z, w)
                        v0.x += rs * elem_lo(r.x); v0.y += rs * elem_hi(r.x); v0.z += rs * elem_lo(r.y); v0.w += rs * elem_hi(r.y);
                        v1.x += rs * elem_lo(r.z); v1.y += rs * elem_hi(r.z); v1.z += rs * elem_lo(r.w); v1.w += rs * elem_hi(r.w);
                    }
                    u32x4_t o;
                    o.x = pack_elem2(v0.x, v0.y); o.y = pack_elem2(v0.z, v0.w);
                    o.z = pack_elem2(v1.x, v1.y); o.w = pack_elem2(v1.z, v1.w);
                    store_pair(ocol, i, o);
                }
            }
            RS_STAMP();      // epilogue issued
#if VMV_RS_RELAX
            if (cend && !late) chunk_end(c == 0 ? SP * (G / 2) : 2 * SP * (G / 2));
#else
            if (cend && !late) chunk_end(SP * (G / 2));
#endif
        }
    }
}

struct RsPlan { int rt, nsplit, cols; };

int rs_policy() {
    // VMV_GEMM_RS (A/B experiments): 1 (default) = this kernel takes the eligible linears, 0 = off
    static int pol = -1;
    if (pol < 0) { const char* e = getenv("VMV_GEMM_RS"); pol = e ? atoi(e) : 1; }
    return pol;
}

int ncu_whole_xcds() {
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        if (n < 8) n = 8;
        ncu = n & ~7;
    }
    return ncu;
}

// rows per wave (RT) and column splits: fewest rounds x block time over the CUs; ties go to the larger row tile (fewer W bytes
// per MAC) and the smaller split
bool rs_plan(const VmvGemmParams& p, int force_rt, RsPlan& best, int ncu) {
    const int K = p.ktot;
    bool found = false;
    double best_cost = 0;
    static int ns_env = -1;
    if (ns_env < 0) { const char* e = getenv("VMV_RS_NSPLIT"); ns_env = e ? atoi(e) : 0; }
    for (int rt = 4; rt >= 2; rt -= 2) {
        if (force_rt && rt != force_rt) continue;
        if ((K == 640 || K == 512) && rt != 2) continue;
        const int bm = 128 * rt;
        const long tm = (p.M + bm - 1) / bm;
        for (int ns = 1; ns <= 8; ns *= 2) {
            if (ns_env > 0 && ns != ns_env) continue;
            if (p.N % (64 * ns)) continue;
            const int cols = p.N / ns;
            if (cols > 5120) continue;
            const long blocks = tm * ns;
            const long rounds = (blocks + ncu - 1) / ncu;
            // block time ~ rows x columns (+ the resident rows' load, which a split repeats)
            const double cost = (double)rounds * ((double)bm * cols + 24.0 * bm * K / 32.0);
            if (!found || cost < best_cost * 0.999) { found = true; best_cost = cost; best = RsPlan{rt, ns, cols}; }
        }
    }
    return found;
}

template <int RT, int KS, int MODE>
int launch_rs(const VmvGemmParams& p, const RsPlan& pl, hipStream_t st) {
    using Cfg = RsCfg<RT, KS>;
    const int tiles_m = (p.M + Cfg::BM - 1) / Cfg::BM;
    static std::atomic<unsigned long long> attr{0};      // one per template instantiation; set once per (kernel, device)
    if (const int rc_attr = vmv_lds_attr_once(attr, reinterpret_cast<const void*>(&gemm_rs_kernel<RT, KS, MODE>), Cfg::LDS_BYTES)) return rc_attr;
    VMV_LAUNCH((gemm_rs_kernel<RT, KS, MODE>), dim3(tiles_m * pl.nsplit), dim3(Cfg::NT), Cfg::LDS_BYTES, st, p, tiles_m, pl.nsplit, pl.cols);
    return vmv_launch_status();
}

template <int RT, int KS>
int launch_rs_mode(const VmvGemmParams& p, const RsPlan& pl, int mode, hipStream_t st) {
    switch (mode) {
        case 0: return launch_rs<RT, KS, 0>(p, pl, st);
        case RS_GEGLU: return launch_rs<RT, KS, RS_GEGLU>(p, pl, st);
        case RS_LN: return launch_rs<RT, KS, RS_LN>(p, pl, st);
        case RS_LN | RS_GEGLU: return launch_rs<RT, KS, RS_LN | RS_GEGLU>(p, pl, st);
        case RS_RES: return launch_rs<RT, KS, RS_RES>(p, pl, st);
        case RS_GN: return launch_rs<RT, KS, RS_GN>(p, pl, st);
        default: return VMV_GLDS_UNSUPPORTED;
    }
}

}  // namespace

// What the row-stationary kernel serves: ONE linear segment covering the whole K = 320 / 640 row, 16-bit staged output, bias /
// GEGLU / residual / folded LayerNorm with in-kernel statistics (colsum + ln_eps; a rowstat from a statistics pass is not
// needed and, with ln_eps given, ignored).
bool vmv_gemm_rs_supported(const VmvGemmParams& p) {
    if (p.nseg != 1 || p.seg[0].mode != VMV_SEG_LINEAR || p.seg[0].k != p.ktot) return false;
    if (p.ktot != 320 && p.ktot != 640 && p.ktot != 512) return false;
    if (p.ksplit > 1 || p.out_fp32 || p.rowvec || p.act != VMV_ACT_NONE || p.wgroup_rows != 0) return false;
    if (p.N % 64) return false;
    const bool geglu = p.epilogue == VMV_EPI_GEGLU;
    const bool ln = p.colsum != nullptr;
    if (ln && !(p.ln_eps > 0.f)) return false;
    if (ln && p.residual) return false;
    if (geglu && p.residual) return false;
    if (p.gn_table) {            // folded GroupNorm: the plain projection only (proj_in), stat groups of whole 16-row tiles, >= a block's rows
        if (ln || geglu || p.residual) return false;
        if (p.gn_rows_per_stat < 512 || (p.gn_rows_per_stat & 15) || (((uintptr_t)p.gn_table) & 15)) return false;
    }
    const int n_out = geglu ? p.N / 2 : p.N;
    if ((p.ldo & 7) || (n_out & 7) || !vmv_aligned16(p.out)) return false;
    if (p.residual && ((p.ldr & 7) || !vmv_aligned16(p.residual))) return false;
    if ((long)(p.M + 512) * p.seg[0].ld * 2 >= (1L << 31) - 65536) return false;
    if ((long)(p.M + 512) * p.ldo * 2 >= (1L << 31) - 65536) return false;
    if (p.residual && (long)(p.M + 512) * p.ldr * 2 >= (1L << 31) - 65536) return false;
    if ((long)p.N * p.ktot * 2 >= (1L << 31) - 65536) return false;
    RsPlan pl;
    return rs_plan(p, 0, pl, 256);
}

// policy: the eligible linears with enough rows to fill the chip (the UNet's two large levels)
bool vmv_gemm_rs_preferred(const VmvGemmParams& p) {
    if (!rs_policy() || !vmv_gemm_rs_supported(p)) return false;
    RsPlan pl;
    if (!rs_plan(p, 0, pl, 256)) return false;
    const long blocks = (long)((p.M + 128 * pl.rt - 1) / (128 * pl.rt)) * pl.nsplit;
    return blocks >= 160;
}

int vmv_gemm_rs_launch(const VmvGemmParams& p, int tile, hipStream_t st) {
    if (!vmv_gemm_rs_supported(p)) return VMV_GLDS_UNSUPPORTED;
    RsPlan pl;
    const int force_rt = tile == VMV_TILE_RS512 ? 4 : tile == VMV_TILE_RS256 ? 2 : 0;
    if (!rs_plan(p, force_rt, pl, ncu_whole_xcds())) return VMV_GLDS_UNSUPPORTED;
    const int mode = (p.epilogue == VMV_EPI_GEGLU ? RS_GEGLU : 0) | (p.colsum ? RS_LN : 0) | (p.residual ? RS_RES : 0) | (p.gn_table ? RS_GN : 0);
    if (p.ktot == 320) return pl.rt == 4 ? launch_rs_mode<4, 10>(p, pl, mode, st) : launch_rs_mode<2, 10>(p, pl, mode, st);
    if (p.ktot == 512) return launch_rs_mode<2, 16>(p, pl, mode, st);      // (the init TemporalTransformer: 8 heads x 64 on a 320-channel level)
    return launch_rs_mode<2, 20>(p, pl, mode, st);
}

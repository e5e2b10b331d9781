// gemm_pglds.hip — PERSISTENT variant of the LDS-DMA bf16 MFMA implicit GEMM (same contract as gemm.hip / vmv.h).
//
// Why: for the short-K linears of the transformer blocks (K = C = 320 / 640: 5-10 chunks of 64 per tile) a
// one-tile-per-block kernel spends most of a tile's life outside its main loop — block launch, per-tile address
// set-up, the first HBM round trip of the ring, and an epilogue that all 8 waves run in lock-step — and with one
// block per CU nothing overlaps any of it (measured: 330-490 TFLOP/s on those shapes, no better with two smaller
// blocks per CU; the MFMA-free and the load-free ablations each run as slowly as the full kernel).  Here ONE block
// per CU walks a list of tiles and the 3-stage ring never drains:
//   * the loader runs two chunks ahead of the MFMAs ACROSS tile boundaries, so tile i+1's first chunks (and its
//     address set-up) are in flight under tile i's last MFMAs and its epilogue;
//   * the epilogue is per-wave: each wave transposes its own 64 x {64,80} accumulator block, 16 rows at a time,
//     through a private slab in the ring slot that tile i has just freed, and writes whole 16-byte lanes
//     (residual prefetched 16 bytes per lane); no block-wide phases, so waves drift apart and one wave's stores
//     overlap another's GELU;
//   * one refill is delayed by one chunk per tile (the slab's slot), two barriers per tile are added.
// Same tile shapes, LDS image, swizzle and fragment schedule as gemm_glds.hip (256 x {128,160}, 8 waves, BK = 64).
// Tile order: item v -> logical tile through the XCD-aware bijection, so the 32 CUs of an XCD work on 32 consecutive
// tiles (same rows of A, neighbouring weight columns) in every round.
#include "gemm_glds_common.h"
#include <cstdlib>
#include <type_traits>

using namespace vmvg;

namespace {

// NWM x 2 waves (NWM = 4: one block per CU, 3-stage ring; NWM = 2: two blocks per CU, 2-stage ring), wave tile
// 16*WM x 16*WN, block tile 16*WM*NWM x 32*WN
template <int NWM, int WM_, int WN>
struct PgCfg {
    static constexpr int NW = 2 * NWM, NT = 64 * NW;
    static constexpr int STAGES = NWM == 4 ? 3 : 2;
    static constexpr int BM = 16 * WM_ * NWM;
    static constexpr int BN = 32 * WN;
    static constexpr int A_BYTES = BM * 128;
    static constexpr int W_BYTES = BN * 128;
    static constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
    static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;
    static constexpr int NAI = BM / (8 * NW);              // A wave-instructions per wave per chunk (8 rows each)
    static constexpr int NWI = (BN / 8 + NW - 1) / NW;
    static constexpr int LPT = NAI + NWI;
    static constexpr int STRIP = 640;                      // per wave: 16*WN bias values + 16*WN column sums (folded LayerNorm)
    static constexpr int STRIPS = STRIP * NW;              // per-wave strips behind the ring
    static constexpr int LDS_LIMIT = (NWM == 4 ? 160 : 80) * 1024;
    static constexpr bool DEDICATED = LDS_BYTES + STRIPS + NW * 2816 <= LDS_LIMIT;     // slabs behind the strips
    static constexpr int LDS_TOTAL = LDS_BYTES + STRIPS + (DEDICATED ? NW * 2816 : 0);
    static_assert(BM % (8 * NW) == 0, "A rows split evenly over the waves");
};

// LNS: the (mean, rstd) of a LayerNorm folded into this GEMM (vmv.h, VmvGemmParams.ln_eps) are accumulated from the A
// fragments the MFMAs consume — the K loop of such a GEMM walks each row completely, a wave's lanes hold output row
// m = frow for both the fragments and the accumulators — instead of being read from a statistics pass.
template <int NWM, int WM, int WN, int ablate, bool IL, bool LNS = false>
__global__ __launch_bounds__(128 * NWM, NWM == 4 ? 1 : 2) void gemm_pglds_kernel(const VmvGemmParams p, const int tiles_n, const int total_steps,
                                                         const int nitems, const int panel_order) {
    VMV_KERNEL_ENTER();
    using Cfg = PgCfg<NWM, WM, WN>;
    constexpr int BN = Cfg::BN;
    constexpr int BM = Cfg::BM;
    constexpr int NW = Cfg::NW;
    constexpr int S = Cfg::STAGES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int G = gridDim.x;
    const int bid = blockIdx.x;

    auto item_tile = [&](int v, int& m0, int& n0) {      // XCD-aware bijection item -> tile (see gemm_glds.hip)
        const int q = nitems >> 3, r = nitems & 7;
        const int xcd = v & 7;
        int idx = v >> 3;
        if (panel_order) {
            // Panels of 8 tile rows (8 * tiles_n consecutive tiles of this XCD's range): CU j of the XCD (idx = 32 round + j)
            // keeps row j % 8 for the whole panel and walks its columns 4 at a time with the three CUs that share the row, so
            // an A tile is fetched from HBM once per panel and re-read from L2 by the following rounds — in plain order every
            // round of an XCD starts on fresh rows and each chunk of every tile waits for an HBM miss.
            const int qx = q + (xcd < r ? 1 : 0);
            const int P = 8 * tiles_n;
            const int pb = (idx / P) * P, l = idx - pb;
            if (pb + P <= qx) idx = pb + (l & 7) * tiles_n + (l >> 3);
        }
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        const int tn = logical % tiles_n;
        m0 = (logical / tiles_n) * BM;
        n0 = tn * BN;
    };

    const int lrow = lane >> 3;
    const int lsw = (lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);    // logical 16-B slot this lane fetches
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, SRD_RECORDS, SRD_FLAGS);

    // ------------------------------------------------------------------ loader state (runs ahead of the MFMAs)
    // (LINEAR segments only — the launcher sends conv / temporal-conv GEMMs, whose K is long anyway, to gemm_glds.hip: a
    //  row's gather state is then just its index, which keeps the loader's registers out of the MFMA loop's way)
    int L_item = bid, L_left = 0;
    int rowm[Cfg::NAI];
    uint32_t wvo[Cfg::NWI], avo[Cfg::NAI];
    int wgrp[Cfg::NWI];
#pragma unroll
    for (int j = 0; j < Cfg::NWI; ++j) {
        int g = j * NW + wave;
        if (g >= BN / 8) g -= NW;
        wgrp[j] = g;
    }
    int s = 0, kc = 0, koff = 0, islot = 0;
    auto enter_segment = [&]() {
#pragma unroll
        for (int i = 0; i < Cfg::NAI; ++i)
            avo[i] = rowm[i] >= 0 ? (uint32_t)(rowm[i] * p.seg[s].ld + lsw * 8) * 2u : OOB;
    };
    auto setup_item = [&](int v) {
        int m0, n0;
        item_tile(v, m0, n0);
#pragma unroll
        for (int i = 0; i < Cfg::NAI; ++i) {
            const int m = m0 + (i * NW + wave) * 8 + lrow;
            rowm[i] = (m < p.M) ? m : -1;
        }
#pragma unroll
        for (int j = 0; j < Cfg::NWI; ++j) {
            const int n = n0 + wgrp[j] * 8 + lrow;
            // (grouped weights, vmv.h: the rows of this tile multiply the weight matrix of group m0 / wgroup_rows)
            const uint32_t wg = p.wgroup_rows > 0 ? (uint32_t)((long)(m0 / p.wgroup_rows) * p.wgroup_stride) * 2u : 0u;
            wvo[j] = (n < p.N) ? (uint32_t)(n * p.ktot + lsw * 8) * 2u + wg : OOB;
        }
        s = 0; kc = 0; koff = 0;
        L_left = total_steps;
        enter_segment();
    };
    // One chunk = LPT wave-instructions ("pieces": NAI of A, then NWI of W).  issue_one() sends them in a burst; the
    // interleaved main loop (IL) sends them one at a time between MFMAs, half a chunk per MFMA phase: the texture path
    // accepts one 1-KB wave-instruction every ~26 cycles per CU (tools/experiments/lds_dma_rate.hip: 91 GB/s per CU with 8
    // waves), so a burst of 8 x LPT instructions after a barrier parks every wave in its issue slot for ~500 cycles with
    // the MFMA pipes idle.
    struct ChunkCtx {
        __amdgpu_buffer_rsrc_t a_rsrc;
        unsigned char* abase;
        unsigned char* wbase;
        uint32_t a_so, w_so;
        bool kall;           // lane's 16 bytes lie inside the segment's K range
    };
    auto chunk_ctx = [&]() -> ChunkCtx {
        const VmvGemmSeg& sg = p.seg[s];
        ChunkCtx c;
        c.a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(sg.src), 0, SRD_RECORDS, SRD_FLAGS);
        c.kall = (kc + BK) <= sg.k || (kc + lsw * 8) < sg.k;
        c.abase = smem + islot * Cfg::STAGE_BYTES + wave * 1024;
        c.wbase = smem + islot * Cfg::STAGE_BYTES + Cfg::A_BYTES;
        c.a_so = (uint32_t)kc * 2u;
        c.w_so = (uint32_t)(koff + kc) * 2u;
        return c;
    };
    auto issue_piece = [&](const ChunkCtx& c, const int q) {      // q: compile-time after unrolling
        if constexpr (ablate != 2) {
            if (q < Cfg::NAI) VMV_BLDS16(c.a_rsrc, c.abase + q * (NW * 1024), c.kall ? avo[q < Cfg::NAI ? q : 0] : OOB, c.a_so);
            else VMV_BLDS16(w_rsrc, c.wbase + wgrp[q >= Cfg::NAI ? q - Cfg::NAI : 0] * 1024,
                            c.kall ? wvo[q >= Cfg::NAI ? q - Cfg::NAI : 0] : OOB, c.w_so);
        }
    };
    auto advance_chunk = [&]() {
        const VmvGemmSeg& sg = p.seg[s];
        islot = islot + 1 == S ? 0 : islot + 1;
        kc += BK;
        --L_left;
        if (kc >= sg.k) {
            koff += sg.k; ++s; kc = 0;
            if (L_left > 0) enter_segment();
        }
        if (L_left == 0) {                   // next tile of this block: its address set-up runs under the current MFMAs
            L_item += G;
            if (L_item < nitems) setup_item(L_item);
        }
    };
    auto issue_one = [&]() {                 // LDS-DMA the loader's next chunk into ring slot `islot` (burst)
        const ChunkCtx c = chunk_ctx();
#pragma unroll
        for (int q = 0; q < Cfg::LPT; ++q) issue_piece(c, q);
        advance_chunk();
    };

    // ------------------------------------------------------------------ MFMA side
    f32x4_t acc[WN][WM];
    float rs1[WM], rs2[WM];                   // LNS: this lane's share of sum x, sum x^2 of output rows frow + 16 i
    auto zero_acc = [&]() {
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int i = 0; i < WM; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if constexpr (LNS) {
#pragma unroll
            for (int i = 0; i < WM; ++i) { rs1[i] = 0.f; rs2[i] = 0.f; }
        }
    };
    zero_acc();
    const int frow = lane & 15;
    const int fgrp = lane >> 4;
    const int fswz = (frow >> 1) & 7;
    auto read_frags = [&](int slot_idx, int kk, elem8_t (&af)[WM], elem8_t (&wf)[WN]) {
        const u32x4_t* a = reinterpret_cast<const u32x4_t*>(smem + slot_idx * Cfg::STAGE_BYTES) + (wave_m * 16 * WM + frow) * 8;
        const u32x4_t* w = reinterpret_cast<const u32x4_t*>(smem + slot_idx * Cfg::STAGE_BYTES + Cfg::A_BYTES) +
                           (wave_n * 16 * WN + frow) * 8;
        const int slot = (kk * 4 + fgrp) ^ fswz;
#pragma unroll
        for (int i = 0; i < WM; ++i) af[i] = __builtin_bit_cast(elem8_t, a[i * 16 * 8 + slot]);
#pragma unroll
        for (int j = 0; j < WN; ++j) wf[j] = __builtin_bit_cast(elem8_t, w[j * 16 * 8 + slot]);
    };
    auto mma = [&](const elem8_t (&af)[WM], const elem8_t (&wf)[WN]) {
        if constexpr (LNS) {
            // + the row sums of the A fragments: 8 dot2c per fragment (sum x against packed ones, sum x^2 against itself), two
            // of them pinned behind each MFMA so that they issue in its shadow (left to the compiler they end up as one
            // block of 48 behind the chunk's MFMAs, with the matrix pipe idle: measured 7-9 % on the L0 / L1 shapes)
            const uint32_t one2 = VMV_ELEM_ONE2;
#pragma unroll
            for (int m = 0; m < WM * WN; ++m) {
                const int j = m / WM, i = m % WM;
                acc[j][i] = VMV_MFMA16(wf[j], af[i], acc[j][i], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int idx = 2 * m + r;
                    if (idx < 8 * WM) {
                        const int fi = idx / 8, e = idx % 8;
                        const u32x4_t u = __builtin_bit_cast(u32x4_t, af[fi]);
                        const uint32_t x = (e >> 1) == 0 ? u.x : (e >> 1) == 1 ? u.y : (e >> 1) == 2 ? u.z : u.w;
                        if (e & 1) elem_dot2c(rs2[fi], x, x); else elem_dot2c(rs1[fi], x, one2);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int i = 0; i < WM; ++i)
                    acc[j][i] = VMV_MFMA16(wf[j], af[i], acc[j][i], 0, 0, 0);
        }
    };

    // One MFMA phase of the interleaved loop: the WM*WN MFMAs on (af, wf), with the WM+WN fragment reads of the NEXT phase
    // (slot_n, kk_n -> afn, wfn) and pieces [Q0, Q1) of the chunk being loaded spread evenly between them.
    constexpr int NM = WM * WN, NRD = WM + WN;
    auto phase = [&](const elem8_t (&af)[WM], const elem8_t (&wf)[WN], elem8_t (&afn)[WM], elem8_t (&wfn)[WN],
                     const int slot_n, const int kk_n, const bool dma, const ChunkCtx& cx, auto q0_tag, auto q1_tag) {
        constexpr int Q0 = decltype(q0_tag)::value, Q1 = decltype(q1_tag)::value, ND = Q1 - Q0;
        const u32x4_t* a = reinterpret_cast<const u32x4_t*>(smem + slot_n * Cfg::STAGE_BYTES) + (wave_m * 16 * WM + frow) * 8;
        const u32x4_t* w = reinterpret_cast<const u32x4_t*>(smem + slot_n * Cfg::STAGE_BYTES + Cfg::A_BYTES) +
                           (wave_n * 16 * WN + frow) * 8;
        const int slot = (kk_n * 4 + fgrp) ^ fswz;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int j = m / WM, i = m % WM;
            if constexpr (ablate != 1) acc[j][i] = VMV_MFMA16(wf[j], af[i], acc[j][i], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < NRD; ++r)
                if (((2 * r + 1) * NM) / (2 * NRD) == m) {
                    if constexpr (ablate != 1) {
                        if (r < WM) afn[r < WM ? r : 0] = __builtin_bit_cast(elem8_t, a[(r < WM ? r : 0) * 16 * 8 + slot]);
                        else wfn[r >= WM ? r - WM : 0] = __builtin_bit_cast(elem8_t, w[(r >= WM ? r - WM : 0) * 16 * 8 + slot]);
                    }
                }
#pragma unroll
            for (int d = 0; d < ND; ++d)
                if (((2 * d + 1) * NM) / (2 * ND) == m) {
                    if (dma) issue_piece(cx, Q0 + d);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ------------------------------------------------------------------ per-wave epilogue through a private LDS slab
    const bool geglu = p.epilogue == VMV_EPI_GEGLU;
    const int N_out = geglu ? p.N / 2 : p.N;
    const bool staged = !p.out_fp32 && (p.ldo & 7) == 0 && (N_out & 7) == 0 && vmv_ptr_aligned16(p.out) &&
                        (!p.residual || ((p.ldr & 7) == 0 && vmv_ptr_aligned16(p.residual)));
    // Split in two so that every global LOAD of the epilogue (bias, residual) is issued in one burst right after the tile's
    // last MFMAs were issued — their latency then overlaps the MFMA drain and the ring wait — and the staging/store part
    // runs without a single memory round trip.  (One load -> use -> store chain per 16-row group, as a naive loop does,
    // costs a full HBM latency per group: measured 11.6k cycles per 256x160 tile without and 21.6k with a residual,
    // against 13k cycles of main loop at K = 320.)  Output / residual go through buffer descriptors: one 32-bit lane
    // offset per 16-byte unit serves all four row groups (the group base is a scalar offset), and out-of-range lanes
    // get the OOB offset — dropped stores, zero loads — instead of 64-bit pointers and exec-masked branches.
    constexpr int OWC_MAX = 16 * WN;
    constexpr int NR_MAX = (16 * (OWC_MAX / 8) + 63) / 64;
    // bias: staged per wave in the 512 B it owns behind the ring (keeps 4*WN registers out of the tile's last MFMAs);
    // residual: two 16-row groups in flight, refilled as soon as a group has been added to its rows
    float* bias_lds = reinterpret_cast<float*>(smem + S * Cfg::STAGE_BYTES + wave * Cfg::STRIP);
    float* csum_lds = bias_lds + 16 * WN;                  // colsum strip (LayerNorm folded into this GEMM: vmv.h)
    u32x4_t resv[2][NR_MAX];                  // residual: two 16-row groups in flight
    u32x4_t sd_prev[NR_MAX];                  // store data of the last 16-row group written (see the epilogue)
#pragma unroll
    for (int r = 0; r < NR_MAX; ++r) sd_prev[r] = u32x4_t{0u, 0u, 0u, 0u};
    const __amdgpu_buffer_rsrc_t bias_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias ? p.N * 4 : 0, SRD_FLAGS);
    const __amdgpu_buffer_rsrc_t csum_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.colsum), 0, p.colsum ? p.N * 4 : 0, SRD_FLAGS);
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, SRD_RECORDS, SRD_FLAGS);
    const __amdgpu_buffer_rsrc_t res_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.residual), 0, SRD_RECORDS, SRD_FLAGS);
    const bool has_res = !LNS && staged && p.residual != nullptr;      // (LNS: no residual — 24 registers the sums need)
    const float res_scale = p.res_scale != 0.f ? p.res_scale : 1.f;
    auto unit_offsets = [&](int m0, int n0, int i, int r, int ld, auto geglu_tag) -> uint32_t {
        // byte offset (relative to the group's scalar base) of 16-byte unit `lane + 64 r` of row group i, or OOB
        constexpr bool GEGLU = decltype(geglu_tag)::value;
        constexpr int OWC = GEGLU ? 8 * WN : 16 * WN;
        constexpr int UW = OWC / 8;
        constexpr int NU = 16 * UW;
        const int unit = lane + 64 * r;
        const int rr = unit / UW, u = unit - rr * UW;
        const int m = m0 + wave_m * 16 * WM + 16 * i + rr;
        const int n = (GEGLU ? n0 / 2 : n0) + wave_n * OWC + u * 8;
        return (unit < NU && m < p.M && n < N_out) ? (uint32_t)(rr * ld + u * 8) * 2u : OOB;
    };
    auto group_base = [&](int m0, int n0, int i, int ld, auto geglu_tag) -> uint32_t {
        constexpr bool GEGLU = decltype(geglu_tag)::value;
        constexpr int OWC = GEGLU ? 8 * WN : 16 * WN;
        const int row0 = m0 + wave_m * 16 * WM + 16 * i;
        const int col0 = (GEGLU ? n0 / 2 : n0) + wave_n * OWC;
        return (uint32_t)__builtin_amdgcn_readfirstlane(row0 * ld + col0) * 2u;
    };
    auto epilogue_prefetch = [&](int m0, int n0, auto geglu_tag) {
        constexpr bool GEGLU = decltype(geglu_tag)::value;
        constexpr int OWC = GEGLU ? 8 * WN : 16 * WN;
        constexpr int NR = (16 * (OWC / 8) + 63) / 64;
        const int nbase = n0 + wave_n * 16 * WN + 4 * fgrp;
        {   // this wave's 16*WN bias values (and, with a folded LayerNorm, column sums) go straight into the wave's LDS strip by
            // 4-byte LDS-DMA: no registers held across the tile's last MFMAs (8 of them made <4,3,5> spill into the epilogue,
            // where every scratch reload waits out the in-order vmcnt queue), no ds_write, and columns >= N read as zero
            // through the descriptor's range check.  They land with the wait_vmcnt<0> that precedes the epilogue.
            constexpr int NB = 16 * WN;
            const int n = n0 + wave_n * NB;
#pragma unroll
            for (int q = 0; q < (NB + 63) / 64; ++q) {
                const int cnt = NB - 64 * q < 64 ? NB - 64 * q : 64;
                if (lane < cnt) {
                    const uint32_t vo = (uint32_t)(n + 64 * q + lane) * 4u;
                    blds4(bias_rsrc, reinterpret_cast<unsigned char*>(bias_lds) + 256 * q, vo, 0);
                    if (LNS || p.rowstat) blds4(csum_rsrc, reinterpret_cast<unsigned char*>(csum_lds) + 256 * q, vo, 0);
                }
            }
        }
        if (has_res) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint32_t sb = group_base(m0, n0, i, p.ldr, geglu_tag);
#pragma unroll
                for (int r = 0; r < NR; ++r)
                    resv[i][r] = __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, unit_offsets(m0, n0, i, r, p.ldr, geglu_tag), sb, 0);
            }
        }
        (void)nbase;
    };
    auto epilogue = [&](int m0, int n0, unsigned char* slot_base, auto geglu_tag) {
        constexpr bool GEGLU = decltype(geglu_tag)::value;
        const int mbase = m0 + wave_m * 16 * WM + frow;
        const int nbase = n0 + wave_n * 16 * WN + 4 * fgrp;
        if (!staged) {
            if constexpr (GEGLU) {
                if constexpr ((WN & 1) == 0) {
#pragma unroll
                    for (int j = 0; j < WN; j += 2)
#pragma unroll
                        for (int i = 0; i < WM; ++i) epilogue_store(p, mbase + 16 * i, nbase + 16 * j, acc[j][i], acc[j + 1][i]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < WN; ++j)
#pragma unroll
                    for (int i = 0; i < WM; ++i) epilogue_store(p, mbase + 16 * i, nbase + 16 * j, acc[j][i], acc[j][i]);
            }
            return;
        }
        constexpr int OWC = GEGLU ? 8 * WN : 16 * WN;         // output columns of this wave
        constexpr int UW = OWC / 8;                           // 16-byte units per slab row
        constexpr int RB = OWC * 2 + 16;                      // slab row pitch (+16 B: spreads the 8-byte writes over banks)
        constexpr int NU = 16 * UW;                           // units per 16-row group
        constexpr int NR = (NU + 63) / 64;
        float2 ms_next = make_float2(0.f, 0.f);
        if constexpr (LNS) {       // lanes frow, frow + 16, + 32, + 48 hold the four k-quarters of row frow's sums
            const float inv_k = 1.0f / (float)p.ktot;
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                float a = rs1[i], b = rs2[i];
                a += __shfl_xor(a, 16); b += __shfl_xor(b, 16);
                a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
                const float mean = a * inv_k;
                const float var = fmaxf(b * inv_k - mean * mean, 0.f);
                rs1[i] = mean; rs2[i] = __builtin_amdgcn_rsqf(var + p.ln_eps);
            }
        } else if (p.rowstat) ms_next = *reinterpret_cast<const float2*>(p.rowstat + (size_t)(mbase < p.M ? mbase : 0) * 2);
        asm volatile("" ::: "memory");
        unsigned char* slab = Cfg::DEDICATED ? smem + Cfg::LDS_BYTES + Cfg::STRIPS + wave * 2816
                                             : slot_base + wave * 4096;                       // 16 rows x RB <= 2816 B
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int m = mbase + 16 * i;
            f32x4_t rv[WN];
            if (p.rowvec) {                                   // (conv1's per-image embedding row: L2 hits, one burst per group)
                const float* rvp = p.rowvec + (size_t)((m < p.M ? m : 0) / p.rowvec_div) * p.rowvec_ld;
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    if (GEGLU && (j & 1)) continue;
                    const int n = nbase + 16 * j;
                    const int no = GEGLU ? (n >> 5) * 16 + (n & 15) : n;
                    rv[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    if (n < p.N) rv[j] = *reinterpret_cast<const f32x4_t*>(rvp + no);
                }
            }
            // Two phases, separated for the scheduler: (A) bias reads + math for every column tile, (B) the slab writes.
            // LayerNorm folded into this GEMM (vmv.h): acc <- rstd[m] * (acc - mean[m] * colsum[n]); the row's two statistics
            // and the columns' sums are L2-resident fp32 (read per 16-row group, only on this path)
            if constexpr (LNS) {
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[j][i] = (acc[j][i] - *reinterpret_cast<const f32x4_t*>(csum_lds + 16 * j + 4 * fgrp) * rs1[i]) * rs2[i];
            } else if (p.rowstat) {       // (column sums from the wave's LDS strip, like the bias; this group's row statistics were
                                   //  requested one group ahead)
                const float2 ms = ms_next;
                if (i + 1 < WM) ms_next = *reinterpret_cast<const float2*>(p.rowstat + (size_t)(m + 16 < p.M ? m + 16 : 0) * 2);
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[j][i] = (acc[j][i] - *reinterpret_cast<const f32x4_t*>(csum_lds + 16 * j + 4 * fgrp) * ms.x) * ms.y;
            }
            u32x2_t packed[WN];
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                if (GEGLU && (j & 1)) continue;
                f32x4_t v = acc[j][i] + *reinterpret_cast<const f32x4_t*>(bias_lds + 16 * j + 4 * fgrp);
                if constexpr (GEGLU) {
                    if constexpr ((WN & 1) == 0) {
                        const f32x4_t g = acc[(j + 1) % WN][i] +
                                          *reinterpret_cast<const f32x4_t*>(bias_lds + 16 * ((j + 1) % WN) + 4 * fgrp);
                        if constexpr (ablate == 3) { v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w; }
                        else { v.x *= gelu_erf_f(g.x); v.y *= gelu_erf_f(g.y); v.z *= gelu_erf_f(g.z); v.w *= gelu_erf_f(g.w); }
                    }
                }
                if (p.rowvec) v += rv[j];
                act_apply(v, p.act);
                packed[j].x = pack_elem2(v.x, v.y); packed[j].y = pack_elem2(v.z, v.w);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                if (GEGLU && (j & 1)) continue;
                const int tc = (GEGLU ? 8 * j : 16 * j) + 4 * fgrp;
                if constexpr (ablate != 8) *reinterpret_cast<u32x2_t*>(slab + frow * RB + tc * 2) = packed[j];
                else if (packed[j].x == 0x12345u) bias_lds[lane] = 1.f;
            }
            // (same wave wrote the slab: LDS executes a wave's instructions in order, no barrier needed — but the COMPILER
            //  must not move the differently-typed reads above the writes, nor the next group's writes above these reads)
            __builtin_amdgcn_s_waitcnt(0xc07f);
            asm volatile("" ::: "memory");
            const uint32_t sb = group_base(m0, n0, i, p.ldo, geglu_tag);
            // All slab reads of the group first, THEN its stores.  The straightforward loop (read unit r -> store it -> read
            // unit r+1 into the same registers) compiled to `buffer_store_dwordx4 v[a:a+3]` directly followed by
            // `ds_read_b128 v[a:a+3]`, and on gfx950 the store intermittently picked up the NEXT unit's first dword in some
            // 16-lane groups (seen as wrong / zero column pairs in rows 7, 9-12 of a group, only when the store path was
            // backed up).  Keeping a store's data registers untouched until the group's reads are done avoids it.
            u32x4_t vout[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int unit = lane + 64 * r;
                const int rr = unit / UW, u = unit - rr * UW;
                vout[r] = u32x4_t{0u, 0u, 0u, 0u};
                if constexpr (ablate != 8) { if (unit < NU) vout[r] = *reinterpret_cast<const u32x4_t*>(slab + rr * RB + u * 16); }
                else vout[r].x = (uint32_t)i;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_sched_barrier(0);
            // Store-data discipline (found with gemm_sglds.hip, where 168 registers make the allocator reuse registers at
            // once): an LDS read that RETURNS into a pending buffer_store's data registers corrupts the store when the store
            // path is backed up.  So store data is always a VALU-written copy (`sd`), never an LDS-read destination, and the
            // previous group's `sd` is kept alive (fake use) until this group's LDS reads — bias strip, slab — have returned.
#pragma unroll
            for (int r = 0; r < NR; ++r)
                if (i > 0) asm volatile("" ::"v"(sd_prev[r]));
            u32x4_t sd[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                u32x4_t v = vout[r];
                if (has_res) {
                    float a[8], b[8];
                    unpack8(v, a); unpack8(resv[i & 1][r], b);
#pragma unroll
                    for (int e = 0; e < 8; ++e) a[e] += res_scale * b[e];
                    v = pack8(a);
                }
                asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                             : "=&v"(sd[r].x), "=&v"(sd[r].y), "=&v"(sd[r].z), "=&v"(sd[r].w)
                             : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
            }
            // The slot just consumed is refilled (group i + 2) BEFORE this group's stores: vmcnt retires in order, so a load
            // issued behind stores can only be waited for together with them (a full write round trip).
            asm volatile("" ::: "memory");
            if (has_res && i + 2 < WM) {
                const uint32_t sr = group_base(m0, n0, i + 2, p.ldr, geglu_tag);
#pragma unroll
                for (int r = 0; r < NR; ++r)
                    resv[i & 1][r] = __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, unit_offsets(m0, n0, i + 2, r, p.ldr, geglu_tag), sr, 0);
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                if constexpr (ablate != 7)
                    __builtin_amdgcn_raw_buffer_store_b128(sd[r], out_rsrc, unit_offsets(m0, n0, i, r, p.ldo, geglu_tag), sb, 0);
                else if (sd[r].x == 0x12345u) bias_lds[lane] = 1.f;          // (keep the value alive)
            }
#pragma unroll
            for (int r = 0; r < NR; ++r) sd_prev[r] = sd[r];
            asm volatile("" ::: "memory");
        }
    };

    // ------------------------------------------------------------------ the flattened chunk pipeline
    const int my_items = bid < nitems ? (nitems - 1 - bid) / G + 1 : 0;
    if (my_items == 0) return;
    const int total = my_items * total_steps;
    setup_item(L_item);
    int issued = 0;
    const int pro = total < S ? total : S;
    for (int i = 0; i < pro; ++i) { issue_one(); ++issued; }
    int consumed = 0;                 // chunks fully multiplied
    int st = 0;                       // ring slot of the next chunk to consume
    bool first = true;
    elem8_t a0[WM], w0[WN], a1[WM], w1[WN];
    bool pending = false;             // IL: the second half of a chunk's pieces is still to be issued
    ChunkCtx cx = chunk_ctx();
    // ablate == 4 (experiments): block 0, wave 0 stamps s_memtime at {tile start, main loop done, epilogue start, epilogue
    // end} into p.workspace (uint64 x 4 per tile)
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(p.workspace);
    int tile_no = 0;
    auto stamp = [&](int k) {
        if constexpr (ablate == 4 || ablate == 7 || ablate == 8) {
            if (bid == 0 && tid == 0 && stamps) stamps[tile_no * 4 + k] = __builtin_readcyclecounter();
        }
    };
    for (int item = bid; item < nitems; item += G) {
        stamp(0);
        // ---- tile prologue: chunk `consumed` must be visible to every wave
        if (first) {
            if (pro == 3) wait_vmcnt<(S == 3 ? 2 : 0) * Cfg::LPT>(); else if (pro == 2) wait_vmcnt<Cfg::LPT>(); else wait_vmcnt<0>();
        }   // (later tiles: everything issued so far was waited for before the previous epilogue)
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();             // also: every wave has left the previous epilogue's slab
        asm volatile("" ::: "memory");
        if (!Cfg::DEDICATED && !first && issued < total) { issue_one(); ++issued; }   // the refill that the epilogue's slab delayed
        if constexpr (ablate != 1) read_frags(st, 0, a0, w0);
        if (!first) {                             // the previous tile's last store data stays alive until these reads returned
            __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
            for (int r = 0; r < NR_MAX; ++r) asm volatile("" ::"v"(sd_prev[r]));
        }
        // chunk consumed+1: with the 3-stage ring it was waited for before the previous epilogue; with the 2-stage ring it
        // was issued right before that epilogue, so exactly the epilogue's stores are younger than it
        bool known_landed = !first && S == 3;
        const int younger_stores = (!first && S == 2 && staged) ? WM * (geglu ? (16 * WN + 63) / 64 : (32 * WN + 63) / 64) : 0;
        int m0, n0;
        item_tile(item, m0, n0);
        if constexpr (IL) {
            constexpr int H0 = Cfg::LPT / 2;
            using QA = std::integral_constant<int, 0>;
            using QB = std::integral_constant<int, H0>;
            using QC = std::integral_constant<int, Cfg::LPT>;
            for (int t = 0; t + 1 < total_steps; ++t) {
                phase(a0, w0, a1, w1, st, 1, pending, cx, QB{}, QC{});           // + second half of the chunk in flight
                if (pending) { advance_chunk(); ++issued; pending = false; }
                const int stn = st + 1 == S ? 0 : st + 1;
                if (!known_landed) {
                    if (issued - consumed >= 3) wait_vmcnt<Cfg::LPT>(); else wait_vmcnt<0>();
                }
                known_landed = false;
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_s_barrier();         // slot st is free, chunk consumed+1 is visible
                asm volatile("" ::: "memory");
                pending = issued < total;
                if (pending) cx = chunk_ctx();
                phase(a1, w1, a0, w0, stn, 0, pending, cx, QA{}, QB{});          // + first half of the next refill
                st = stn;
                ++consumed;
            }
            if (pending) {                            // tile boundary: the rest of the refill goes out in one piece
#pragma unroll
                for (int q = H0; q < Cfg::LPT; ++q) issue_piece(cx, q);
                advance_chunk(); ++issued; pending = false;
            }
        } else
        for (int t = 0; t + 1 < total_steps; ++t) {
            if constexpr (ablate != 1) {
                read_frags(st, 1, a1, w1);
                __builtin_amdgcn_sched_barrier(0);
                mma(a0, w0);
            }
            const int stn = st + 1 == S ? 0 : st + 1;
            if (!known_landed) {                  // chunk consumed+1 landed (mine); one younger chunk may stay in flight
                if (S == 3 && issued - consumed >= 3) wait_vmcnt<(S == 3 ? 1 : 0) * Cfg::LPT>();
                else if (S == 2 && t == 0 && younger_stores > 0) wait_vmcnt_rt(younger_stores);
                else wait_vmcnt<0>();
            }
            known_landed = false;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();         // slot st is free, chunk consumed+1 is visible
            asm volatile("" ::: "memory");
            if constexpr (ablate != 1) {
                read_frags(stn, 0, a0, w0);
                __builtin_amdgcn_sched_barrier(0);
                mma(a1, w1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (issued < total) { issue_one(); ++issued; }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            st = stn;
            ++consumed;
        }
        // bias + residual start their round trip under the tile's last 2 x (WM*WN) MFMAs (issuing them earlier, inside the
        // loop, measurably slowed the loop itself; later, after the MFMAs, exposes ~4k cycles of HBM latency per tile)
        asm volatile("" ::: "memory");
        if (geglu) epilogue_prefetch(m0, n0, std::true_type{}); else epilogue_prefetch(m0, n0, std::false_type{});
        asm volatile("" ::: "memory");
        if constexpr (ablate != 1) {              // last chunk of the tile
            read_frags(st, 1, a1, w1);
            mma(a0, w0);
            mma(a1, w1);
        }
        ++consumed;
        stamp(1);
        unsigned char* slab_slot = smem + st * Cfg::STAGE_BYTES;
        st = st + 1 == S ? 0 : st + 1;
        // ---- everything in flight (the next tile's first chunks, issued >= 1 MFMA batch ago) lands; then all waves have
        //      finished reading slot `st-1`, which becomes the epilogue's slab space
        wait_vmcnt<0>();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (Cfg::DEDICATED && issued < total) { issue_one(); ++issued; }     // slabs are outside the ring: refill at once
        asm volatile("" ::: "memory");
        stamp(2);
        if (geglu) epilogue(m0, n0, slab_slot, std::true_type{}); else epilogue(m0, n0, slab_slot, std::false_type{});
        zero_acc();
        first = false;
        stamp(3);
        ++tile_no;
    }
}

template <int NWM, int WM, int WN>
int launch_pglds(const VmvGemmParams& p, int total_steps, hipStream_t st) {
    using Cfg = PgCfg<NWM, WM, WN>;
    static_assert(Cfg::LPT >= 6 && Cfg::LPT <= 9 && (Cfg::STAGES == 2 || 2 * Cfg::LPT <= 14), "wait_vmcnt literals");
    static_assert(Cfg::DEDICATED || (Cfg::NW * 4096 <= Cfg::STAGE_BYTES), "per-wave slabs fit in one ring slot");
    static_assert(16 * (16 * WN * 2 + 16) <= 2816 && 16 * WN * 4 <= 512 && Cfg::LDS_TOTAL <= Cfg::LDS_LIMIT, "LDS budget");
    const int tiles_m = (p.M + Cfg::BM - 1) / Cfg::BM;
    const int tiles_n = (p.N + Cfg::BN - 1) / Cfg::BN;
    const int nitems = tiles_m * tiles_n;
    static int ncu = 0;
    int ncu_eff = ncu;
    if (ncu == 0) {
        int dev = 0, n = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) {
            if (!vmv_dry_run) return (int)e;
            n = 256;                                 // (vmv_gemm_validate without a device: the MI355X count, not cached)
        }
        if (n < 8) n = 8;
        ncu_eff = n & ~7;                            // whole XCD groups: item & 7 == block & 7 in every round
        if (e == hipSuccess) ncu = ncu_eff;
    }
    static int ablate = -1;
    if (ablate < 0) { const char* e = getenv("VMV_GEMM_ABLATE"); ablate = e ? atoi(e) : 0; }
    static int order_env = -1;
    if (order_env < 0) { const char* e = getenv("VMV_GEMM_ORDER"); order_env = e ? atoi(e) : 0; }
    const int slots = ncu_eff * (NWM == 4 ? 1 : 2);
    const int G = nitems < slots ? nitems : slots;
    dim3 grid(G, 1, 1);
    const int order = (order_env == 1 && G == 256 && NWM == 4) ? 1 : 0;       // (the panel map assumes 32 single-block CUs per XCD)
    static int il_env = -1;
    if (il_env < 0) { const char* e = getenv("VMV_GEMM_IL"); il_env = e ? atoi(e) : 0; }
    auto go_il = [&](auto tag, auto il_tag) -> int {
        constexpr int AB = decltype(tag)::value;
        constexpr bool IL = decltype(il_tag)::value;
        static std::atomic<unsigned long long> attr_set{0};
        if (const int rc_attr = vmv_lds_attr_once(attr_set, reinterpret_cast<const void*>(&gemm_pglds_kernel<NWM, WM, WN, AB, IL>), Cfg::LDS_TOTAL)) return rc_attr;
        VMV_LAUNCH((gemm_pglds_kernel<NWM, WM, WN, AB, IL>), grid, dim3(Cfg::NT), Cfg::LDS_TOTAL, st, p, tiles_n, total_steps,
                           nitems, order);
        return VMV_OK;
    };
    auto go = [&](auto tag) -> int {
#if defined(VMV_EXPERIMENTS)
        if constexpr (NWM == 4) { if (il_env == 1) return go_il(tag, std::true_type{}); }
#endif
        return go_il(tag, std::false_type{});
    };
    int rc;
    if (vmv_gemm_ln_inline(p)) {               // row statistics in the main loop: the one-block-per-CU configurations only
        if constexpr (NWM == 4) {
            static std::atomic<unsigned long long> attr_set_lns{0};
            if (const int rc_attr = vmv_lds_attr_once(attr_set_lns, reinterpret_cast<const void*>(&gemm_pglds_kernel<NWM, WM, WN, 0, false, true>), Cfg::LDS_TOTAL)) return rc_attr;
            VMV_LAUNCH((gemm_pglds_kernel<NWM, WM, WN, 0, false, true>), grid, dim3(Cfg::NT), Cfg::LDS_TOTAL, st, p, tiles_n,
                               total_steps, nitems, order);
            return vmv_launch_status();
        } else {
            return VMV_EINVAL;
        }
    }
#if defined(VMV_EXPERIMENTS)       // (the ablation / stamp instantiations are not in the production library)
    switch (ablate) {
        case 1: rc = go(std::integral_constant<int, 1>{}); break;
        case 2: rc = go(std::integral_constant<int, 2>{}); break;
        case 3: rc = go(std::integral_constant<int, 3>{}); break;
        case 4: rc = go(std::integral_constant<int, 4>{}); break;
        case 7: rc = go(std::integral_constant<int, 7>{}); break;
        case 8: rc = go(std::integral_constant<int, 8>{}); break;
        default: rc = go(std::integral_constant<int, 0>{}); break;
    }
#else
    rc = go(std::integral_constant<int, 0>{});
#endif
    if (rc != VMV_OK) return rc;
    return vmv_launch_status();
}

}  // namespace

// Called by vmv_gemm (gemm.hip) after argument validation; split-K shapes stay on the non-persistent kernels.
bool vmv_gemm_pglds_supported(const VmvGemmParams& p) {
    if (p.ksplit > 1) return false;
    for (int i = 0; i < p.nseg; ++i)
        if (p.seg[i].mode != VMV_SEG_LINEAR) return false;
    long maxrows = p.M;
    if (p.OH > 0) { const long src_rows = (long)(p.M / (p.OH * p.OW) + 1) * p.IH * p.IW; if (src_rows > maxrows) maxrows = src_rows; }
    for (int i = 0; i < p.nseg; ++i)
        if (maxrows * (long)p.seg[i].ld * 2 >= (1L << 31) - 65536) return false;
    if ((long)p.N * p.ktot * 2 >= (1L << 31) - 65536) return false;
    // the epilogue addresses out / residual through buffer descriptors too (32-bit byte offsets)
    if ((long)(p.M + 256) * p.ldo * (p.out_fp32 ? 4 : 2) >= (1L << 31) - 65536) return false;
    if (p.residual && (long)(p.M + 256) * p.ldr * 2 >= (1L << 31) - 65536) return false;
    return true;
}

int vmv_gemm_pglds_launch(const VmvGemmParams& p, int total_steps, int tile, hipStream_t st) {
    if (!vmv_gemm_pglds_supported(p)) return VMV_GLDS_UNSUPPORTED;
    if (tile == VMV_TILE_P256x128) return launch_pglds<4, 4, 4>(p, total_steps, st);
    if (tile == VMV_TILE_Q128x128) return launch_pglds<2, 4, 4>(p, total_steps, st);
    if (tile == VMV_TILE_Q96x160) {
        if (p.epilogue == VMV_EPI_GEGLU) return VMV_EINVAL;
        return launch_pglds<2, 3, 5>(p, total_steps, st);
    }
    if (tile == VMV_TILE_P256x160) {
        if (p.epilogue == VMV_EPI_GEGLU) return VMV_EINVAL;
        return launch_pglds<4, 3, 5>(p, total_steps, st);  // 192 x 160: the 160-wide wave tile needs 20 fewer accumulators
    }
    return VMV_EINVAL;
}

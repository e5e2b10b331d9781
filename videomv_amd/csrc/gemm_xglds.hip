// gemm_xglds.hip — 8-wave LDS-DMA implicit GEMM with WIDE wave tiles and a four-stage ring of 32-deep chunks (same contract
// as gemm.hip / vmv.h).
//
// Why: per 64-deep chunk of its 256 x 160 tile gemm_glds.hip's eight 64 x 80 wave tiles pull 147 KB of fragments out of LDS
// (ds_read_b128: 256 B/clk at best) while the LDS-DMA writes 52 KB into it, and the CU's DMA path needs ~1350 cycles for the
// 52 1-KB wave-instructions — against 1280 cycles of MFMAs per SIMD.  The long-K convolutions therefore sit at ~43 % of peak
// with the matrix pipes, the DMA path and the LDS port all busy.  Here the block tile is 256 x 320 (N = 320 / 640 / 1280 are
// all multiples) and a wave owns 64 x 160: per MAC 22 % fewer fragment bytes and 31 % fewer DMA bytes; per barrier interval
// (one 32-deep chunk: 1280 MFMA cycles per SIMD) 115 KB of fragment reads, 36 KB of DMA (~940 cycles of the DMA path).
// 160 accumulators per lane leave room for ONE double-buffered half-chunk of fragments: a chunk is two phases of 20 MFMAs
// (column half h), the fragment reads of the next phase (5 W fragments, plus the 4 A fragments when the chunk changes) are
// issued before the MFMAs of the current one; the second wave of the SIMD covers what latency remains.  A 72-KB stage per
// 64-deep chunk would allow two stages only, hence 32-deep chunks (36 KB) and FOUR stages: chunk t + 4 goes out behind the
// block barrier of step t.  LDS rows are 64 B with a 4-entry slot swizzle (see the loader).  Zero fill by
// descriptor and the K-segment walk are those of gemm_glds.hip; no split-K.  Round 4: the 256 x 256 tile also carries the two
// epilogues of the transformer linears (template EPI) — the folded LayerNorm (rowstat + colsum: rstd (acc - mean colsum) + b') and
// GEGLU (x | gate column tiles are adjacent accumulators of one lane) — for the K = 1280 level, where the row-stationary kernel does
// not apply (rows do not fit registers) and the 256 x 128 persistent tiles ran at 780-870 TFLOP/s against 1050-1100 for this one
// (hipBLASLt on the same box: 1090-1130, profiles/r4_vendor_yardstick.tsv).
// (tools/experiments/gemm_wglds.hip is the 4-wave / 512-register sibling that lost to LDS-DMA issue stalls.)
#include "gemm_glds_common.h"
#include <cstdlib>
#include <type_traits>

using namespace vmvg;

namespace {

#ifndef VMV_XGLDS_VARIANT
#define VMV_XGLDS_VARIANT 0    // experiments: 1 LDS-DMA issue before the chunk's last MFMA phase, 2 s_setprio 1 around MFMA phases, 3 both
#endif
#ifndef VMV_XGLDS_ABLATE
#define VMV_XGLDS_ABLATE 0     // experiments: 1 no MFMAs, 2 no LDS-DMA after the prologue, 3 no fragment reads, 4 no block barriers
#endif
constexpr int WBK = 32;                 // this kernel's K chunk: ONE k-step of v_mfma_f32_16x16x32 (see above)

// WNV = the wave grid: 2 = 4 x 2 waves (the 256-row tiles); 1 = 8 x 1 (round 6: a 512 x 128 tile of 64 x 128 wave tiles for the N = 128
// convolutions of the VAE's first level, whose 64 x 64 wave tiles in every other kernel read 0.5 fragments per MFMA and sit at
// 670 TFLOP/s whatever the tile: 0.375 here); 4 = 4 x 1 waves in a 256-THREAD block, 256 x 128 tile, three-stage ring (72 KB) and TWO
// blocks per CU (round 6 experiment for the K = 1280 linears: still 2 waves per SIMD and 64 x 128 wave tiles, but the two blocks of a CU
// are independent, so one's fill / epilogue runs under the other's main loop — the 14.7 us per 47.6-us tile the one-block form exposes.
// MEASURED: correct, and slower everywhere — qkv L2 644 vs 754 TFLOP/s, GEGLU L2 756 vs 818, FF-down L2 608 vs 977: a 256 x 128 block
// moves 50 % more LDS-DMA bytes per flop than the 256 x 256 one and its steady state is DMA-bound at ~620 TFLOP/s; built with
// EXPERIMENTS=1 only, profiles/r6_y256_bench.log).
template <int NH, int WH, int WNV = 2>
struct WgCfg {
    static constexpr int WM = 4, WN = NH * WH;              // 16-row / 16-column MFMA tiles per wave
    static constexpr int NW = WNV == 4 ? 4 : 8, NT = 64 * NW;          // waves per block: WMV along M x WNW along N
    static constexpr int WNW = WNV == 2 ? 2 : 1;
    static constexpr int WMV = NW / WNW;
    static constexpr int BPC = WNV == 4 ? 2 : 1;            // blocks per CU
    static constexpr int BM = 16 * WM * WMV, BN = 16 * WN * WNW;
    static constexpr int A_BYTES = BM * 64, W_BYTES = BN * 64;         // rows of 32 elements = 64 B
    static constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
#ifndef VMV_XGLDS_STAGES
#define VMV_XGLDS_STAGES 4      // (experiments: 3 = the depth a persistent form with per-wave epilogue slabs could afford)
#endif
    static constexpr int STAGES = WNV == 4 ? 3 : VMV_XGLDS_STAGES;
    static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;
    static constexpr int NAI = BM / (16 * NW);              // A wave-instructions per wave per chunk (16 rows x 64 B each) = 2 (4 at BM = 512)
    static constexpr int WGROUPS = BN / 16;                 // 16-row groups of W per chunk (20 at BN = 320)
    static constexpr int NWI = WGROUPS / NW;                // W wave-instructions of EVERY wave per chunk ...
    static constexpr int NWX = WGROUPS % NW;                // ... and one more for the waves < NWX
    static constexpr int LPT = NAI + NWI;                   // loads per lane per chunk (waves < NWX: LPT + 1)
    static constexpr int HALF_ROWS = 128;                   // epilogue staging: 128 rows (two wave rows) at a time
    static constexpr int XG_MAXG = 8;                       // row groups (rowvec rows) one tile's rows may span
    static_assert(WNV == 1 || WNV == 2 || WNV == 4, "wave grid");
    static_assert(HALF_ROWS * (BN * 2 + 16) + XG_MAXG * BN * 4 + BN * 4 <= LDS_BYTES && BPC * LDS_BYTES <= 160 * 1024, "LDS budget");
};

constexpr int XE_GEGLU = 1, XE_LN = 2;       // EPI bits

// SK (round 4): split-K — blockIdx.y owns `nsteps_arg` chunks of the K walk starting at blockIdx.y * nsteps_arg (the last split takes
// what is left; the launcher keeps every split even and >= STAGES) and writes its raw fp32 accumulators to slab blockIdx.y of
// p.workspace; gemm_splitk_reduce (gemm.hip) sums the slabs and runs the epilogue.  For the grids that cannot fill the chip with
// 256-row wide tiles on their own (the fourth UNet level, a frame-parallel rank's M / 8 rows) without falling back to 128-row tiles.
template <int NH, int WH, int EPI = 0, bool SK = false, int WNV = 2>
__global__ __launch_bounds__(WNV == 4 ? 256 : 512, WNV == 4 ? 2 : 1) void gemm_xglds_kernel(const VmvGemmParams p, const int tiles_m, const int tiles_n, const int nsteps_arg,
                                                            const int nsteps_total, const int gm, const int tapmajor) {
    VMV_KERNEL_ENTER();
    using Cfg = WgCfg<NH, WH, WNV>;
    constexpr int WM = Cfg::WM, WN = Cfg::WN, NW = Cfg::NW, BM = Cfg::BM, BN = Cfg::BN, S = Cfg::STAGES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = WNV == 2 ? wave >> 1 : wave, wave_n = WNV == 2 ? wave & 1 : 0;

    // ---- XCD-aware tile mapping (bijective, as gemm_glds.hip)
    const int nblk = tiles_m * tiles_n;
    int logical;
    {
        const int bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // Round 6 — which tiles run TOGETHER on an XCD decides what its 4-MB L2 can share.  With the N tiles of one row tile adjacent, the
    // 32 CUs of an XCD work on ONE row tile and 32 different W slices: A is shared, every W slice streams in from the fabric once PER
    // ROW TILE (profiles/r6_gemm_traffic_by_kernel.tsv: the 7680 x 10240 x 1280 GEGLU fetches 834 MB for 46 MB of operands = W x 30 row
    // tiles).  Grouped order (gm > 1): consecutive logical ids walk gm row tiles, then the next N tile — 32 concurrent blocks cover
    // gm row tiles x 32 / gm column tiles, so a W slice is fetched once per gm row tiles and an A slice once per 32 / gm column tiles.
    int tm_, tn_;
    if (gm > 1) {
        const int gsz = gm * tiles_n, g = logical / gsz, first = g * gm;
        const int gmh = tiles_m - first < gm ? tiles_m - first : gm;
        const int rem = logical - g * gsz;
        tn_ = rem / gmh; tm_ = first + (rem - tn_ * gmh);
    } else {
        tm_ = logical / tiles_n; tn_ = logical - tm_ * tiles_n;
    }
    const int m0 = tm_ * BM, n0 = tn_ * BN;

    // ---- loader.  A wave instruction covers 16 rows x 64 B; lane -> (row in group = lane >> 2, physical 16-B slot = lane & 3).
    //      Row r keeps logical k-slot s at s ^ T[(r >> 2) & 3], T = {0, 2, 3, 1}: ds_read_b128 serves a wave in four groups
    //      of 16 lanes that are NOT contiguous — {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 — i.e. fragment
    //      rows 0-3 and 12-15 with k-slot f, rows 4-11 with k-slot f + 1 (or the other way round); with this T the four
    //      rows of a group that share r % 4 land in four different 16-byte bank units (the plain (r >> 2) & 3 makes them
    //      collide in pairs).  Applied to the SOURCE address (the LDS image of a DMA is lane-linear); for the rows of this
    //      lane (16 g + (lane >> 2)) the index is (lane >> 4) & 3.
    const int lrow = lane >> 2;
    const int lsw = (lane & 3) ^ ((0x78 >> (2 * ((lane >> 4) & 3))) & 3);
    // weights: this lane's row of W group j is n0 + (j NW + wave) 16 + lrow — linear in j, so ONE offset register; rows >= N
    // fall outside the descriptor and read as zero
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, (uint32_t)p.N * (uint32_t)p.ktot * 2u, SRD_FLAGS);
    const uint32_t wvo0 = (uint32_t)((n0 + wave * 16 + lrow) * p.ktot + lsw * 8) * 2u;
    const uint32_t wstride = (uint32_t)(NW * 16 * p.ktot) * 2u;

    // ---- K walk: RUNS of segments, walked chunk-major (round 6).  A run is a maximal sequence of consecutive segments that differ only
    //      in their tap — the nine of a 3 x 3 convolution, the three of a temporal one; every other segment (linear, strided /
    //      up-sampling taps, everything under split-K) is a run of one.  Inside a run the walk takes the 32 channels of chunk c for EVERY
    //      tap, then chunk c + 1.  The segment-major walk of rounds 2-5 re-read a tile's rows (6 image rows for 4) once per tap with all
    //      the tile's other channels in between: 327 KB per tile and tap at the first level, x 32 CUs per XCD = 10 MB of reuse distance
    //      against a 4-MB L2, so every tap came in over the fabric (profiles/r6_gemm_traffic_by_kernel.tsv: 1.4-2.2 GB fetched per
    //      launch for 160-320 MB of operands).  Chunk-major the taps of a chunk touch 24 KB per tile: eight of nine hit the L2.  W needs
    //      no repacking — its K order stays (tap, channel); the walk takes column offset tap * C + 32 c.
    //      Per lane and A row: ONE byte offset (the row's tap-(0, 0) source position; for a run of one the tap's own position) and a
    //      validity bit per tap (image / frame borders, rows >= M), computed once per run; per chunk one scalar delta and a select.  The
    //      row's (image, y, x) is recomputed from m at every run start — two integer divisions per row once per source tensor — instead
    //      of living in four registers beside 160 accumulators (the 256 x 320 form has none to spare; the asm below keeps the compiler
    //      from hoisting the divisions out of the K loop and back into registers).
    int s = 0, koff = 0;
    int run_len = 1, run_t = 0, run_nch = 0, run_c = 0;
    int nsteps = nsteps_arg;
    int skip0 = 0;
    if constexpr (SK) {           // fast-forward the K walk to this split's first chunk
        const int step_begin = (int)blockIdx.y * nsteps_arg;
        nsteps = nsteps_total - step_begin < nsteps_arg ? nsteps_total - step_begin : nsteps_arg;
        int skip = step_begin;
        while (s < p.nseg) {
            const int nch = (p.seg[s].k + WBK - 1) / WBK;
            if (skip < nch) { skip0 = skip; break; }
            skip -= nch; koff += p.seg[s].k; ++s;
        }
    }
    uint32_t avo[Cfg::NAI];
    uint32_t rmask[(Cfg::NAI + 1) / 2];      // tap-validity bits of the lane's NAI rows, 16 per row, two rows per register
    auto seg_same = [&](const VmvGemmSeg& a, const VmvGemmSeg& b) { return a.src == b.src && a.ld == b.ld && a.k == b.k && a.mode == b.mode; };
    auto start_run = [&]() {
        run_len = 1; run_t = 0; run_c = 0;
        const VmvGemmSeg& s0 = p.seg[s];
        run_nch = (s0.k + WBK - 1) / WBK;
        const int mode = s0.mode, ld = s0.ld;
        const bool tappable = (mode == VMV_SEG_SPATIAL && p.stride == 1 && p.ups == 0) || mode == VMV_SEG_TEMPORAL;
        if (!SK && tapmajor && tappable)
            while (s + run_len < p.nseg && run_len < 15 && seg_same(p.seg[s + run_len], s0)) ++run_len;
        // (segment fields are read in wave-uniform control flow: inside per-row branches the compiler copied the whole kernel-argument
        //  segment table to scratch to index it)
        const bool single = run_len == 1;
        // (the segment table is read in wave-uniform code only — tap loop outside, rows inside: indexed from inside the per-row code the
        //  compiler copies the whole kernel-argument table to scratch)
        int nb[Cfg::NAI], oy[Cfg::NAI], ox[Cfg::NAI];      // spatial: image base row n IH IW, output (y, x); temporal: nb = frame index
        uint32_t mask[Cfg::NAI];
        bool inm[Cfg::NAI];
#pragma unroll
        for (int i = 0; i < Cfg::NAI; ++i) {
            int m = m0 + (i * NW + wave) * 16 + lrow;
            asm volatile("" : "+v"(m));                 // not loop-invariant as far as the compiler can tell: see above
            nb[i] = 0; oy[i] = 0; ox[i] = 0; mask[i] = 0;
            if (mode == VMV_SEG_SPATIAL) {
                const int hw = p.OH * p.OW;
                const int n = m / hw, rem = m - n * hw;
                oy[i] = rem / p.OW; ox[i] = rem - oy[i] * p.OW;
                nb[i] = n * p.IH * p.IW;
            } else if (mode == VMV_SEG_TEMPORAL) {
                nb[i] = (m / p.P) % p.F;
            }
            inm[i] = m < p.M;
            int base;
            if (mode == VMV_SEG_LINEAR) {
                base = m * ld; mask[i] = inm[i] ? 1u : 0u;
            } else if (mode == VMV_SEG_SPATIAL) {
                if (single) {                           // the tap's own position (strided / nearest-x2 gathers included)
                    const int iy = oy[i] * p.stride + s0.d0, ix = ox[i] * p.stride + s0.d1;
                    const int VH = p.IH << p.ups, VW = p.IW << p.ups;
                    const bool ok = inm[i] && iy >= 0 && iy < VH && ix >= 0 && ix < VW;
                    base = ok ? (nb[i] + (iy >> p.ups) * p.IW + (ix >> p.ups)) * ld : 0;
                    mask[i] = ok ? 1u : 0u;
                } else {
                    base = (nb[i] + oy[i] * p.IW + ox[i]) * ld;
                }
            } else {
                if (single) {
                    const int f = nb[i] + s0.d0;
                    const bool ok = inm[i] && f >= 0 && f < p.F;
                    base = ok ? (m + s0.d0 * p.P) * ld : 0;
                    mask[i] = ok ? 1u : 0u;
                } else {
                    base = m * ld;
                }
            }
            avo[i] = (uint32_t)(base + lsw * 8) * 2u;
        }
        if (!single) {
            for (int t = 0; t < run_len; ++t) {
                const int d0 = p.seg[s + t].d0, d1 = p.seg[s + t].d1;
#pragma unroll
                for (int i = 0; i < Cfg::NAI; ++i) {
                    bool ok;
                    if (mode == VMV_SEG_SPATIAL) {
                        const int iy = oy[i] + d0, ix = ox[i] + d1;
                        ok = iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
                    } else {
                        const int f = nb[i] + d0;
                        ok = f >= 0 && f < p.F;
                    }
                    if (ok && inm[i]) mask[i] |= 1u << t;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < (Cfg::NAI + 1) / 2; ++i) rmask[i] = 0;
#pragma unroll
        for (int i = 0; i < Cfg::NAI; ++i) rmask[i >> 1] |= mask[i] << (16 * (i & 1));
    };
    start_run();
    if constexpr (SK) run_c = skip0;
    auto issue = [&](int stage) {           // LDS-DMA one chunk into ring slot `stage`, then advance the K walk
        unsigned char* abase = smem + stage * Cfg::STAGE_BYTES + wave * 1024;
        unsigned char* wbase = smem + stage * Cfg::STAGE_BYTES + Cfg::A_BYTES + wave * 1024;
        const VmvGemmSeg& sg = p.seg[s + run_t];
        const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(sg.src), 0, SRD_RECORDS, SRD_FLAGS);
        const int kc = run_c * WBK;
        const bool kvalid = (kc + WBK) <= sg.k || (kc + lsw * 8) < sg.k;      // k tail of a segment: zero fill
        // the tap's distance from the run's base position, elements, wave-uniform.  It may be negative: it goes into the 32-bit lane
        // offset (a valid tap's sum is a non-negative offset inside the tensor), not into the unsigned scalar offset
        int delta = 0;
        if (run_len > 1) delta = sg.mode == VMV_SEG_SPATIAL ? (sg.d0 * p.IW + sg.d1) * sg.ld : sg.d0 * p.P * sg.ld;
        const uint32_t a_so = (uint32_t)kc * 2u, w_so = (uint32_t)(koff + run_t * sg.k + kc) * 2u, d2 = (uint32_t)(delta * 2);
#pragma unroll
        for (int i = 0; i < Cfg::NAI; ++i)
            VMV_BLDS16(a_rsrc, abase + i * (NW * 1024), (kvalid && ((rmask[i >> 1] >> (16 * (i & 1) + run_t)) & 1u)) ? avo[i] + d2 : OOB, a_so);
#pragma unroll
        for (int j = 0; j < Cfg::NWI; ++j) VMV_BLDS16(w_rsrc, wbase + j * (NW * 1024), kvalid ? wvo0 + (uint32_t)j * wstride : OOB, w_so);
        if (Cfg::NWX > 0 && wave < Cfg::NWX)
            VMV_BLDS16(w_rsrc, wbase + Cfg::NWI * (NW * 1024), kvalid ? wvo0 + (uint32_t)Cfg::NWI * wstride : OOB, w_so);
        if (++run_t == run_len) {
            run_t = 0;
            if (++run_c == run_nch) {
                koff += run_len * sg.k; s += run_len;
                if (s < p.nseg) start_run();
            }
        }
    };

    // ---- MFMA side
    f32x4_t acc[WN][WM];
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int i = 0; i < WM; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fgrp = lane >> 4;
    const int fslot = fgrp ^ ((0x78 >> (2 * ((frow >> 2) & 3))) & 3);     // physical 16-B slot of this lane's k-slice in every fragment row
    elem8_t af[2][WM], wf[2][WH];
    auto read_a = [&](int slot_idx, elem8_t (&a)[WM]) {
        const u32x4_t* ap = reinterpret_cast<const u32x4_t*>(smem + slot_idx * Cfg::STAGE_BYTES) + (wave_m * 16 * WM + frow) * 4 + fslot;
#pragma unroll
        for (int i = 0; i < WM; ++i) a[i] = __builtin_bit_cast(elem8_t, ap[i * 16 * 4]);
    };
    auto read_w = [&](int slot_idx, int h, elem8_t (&w)[WH]) {
        const u32x4_t* wp = reinterpret_cast<const u32x4_t*>(smem + slot_idx * Cfg::STAGE_BYTES + Cfg::A_BYTES) +
                            (wave_n * 16 * WN + h * 16 * WH + frow) * 4 + fslot;
#pragma unroll
        for (int j = 0; j < WH; ++j) w[j] = __builtin_bit_cast(elem8_t, wp[j * 16 * 4]);
    };
    // A chunk is NH phases of WM * WH MFMAs (column half h).  Phase (t, h) multiplies af[t & 1] with wf[(t NH + h) & 1]; the
    // fragments of the NEXT phase are read before its MFMAs are issued.  The chunk parity is a compile-time value: the loop
    // body covers two chunks.
    auto mma_phase = [&](auto par_tag, auto h_tag) {
        constexpr int par = decltype(par_tag)::value, h = decltype(h_tag)::value;
        constexpr int ws = (par * NH + h) & 1;
#if VMV_XGLDS_VARIANT >= 2
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int j = 0; j < WH; ++j)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#if VMV_XGLDS_ABLATE == 1
                { if (i == 0 && j == 0) acc[h * WH][0][0] += (float)wf[ws][0][0] + (float)af[par][0][0]; }
#else
                acc[h * WH + j][i] = VMV_MFMA16(wf[ws][j], af[par][i], acc[h * WH + j][i], 0, 0, 0);
#endif
#if VMV_XGLDS_VARIANT >= 2
        __builtin_amdgcn_s_setprio(0);
#endif
    };
    auto prefetch_phase = [&](int slot_idx, auto par_tag, auto h_tag) {     // the fragment reads phase (par, h) needs
        constexpr int par = decltype(par_tag)::value, h = decltype(h_tag)::value;
#if VMV_XGLDS_ABLATE == 3
        (void)slot_idx;
#else
        if constexpr (h == 0) read_a(slot_idx, af[par]);
        read_w(slot_idx, h, wf[(par * NH + h) & 1]);
#endif
    };

    // (the launcher guarantees an even nsteps >= S: the loop body is two chunks, the fragment-set parity a compile-time value)
    int issued = 0;
#pragma unroll
    for (int i = 0; i < S; ++i) { issue(i); ++issued; }
    const bool xw = Cfg::NWX > 0 && wave < Cfg::NWX;          // this wave issues LPT + 1 loads per chunk (uniform)
    if (xw) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 1) * (Cfg::LPT + 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 1) * Cfg::LPT) : "memory");      // chunk 0 landed (mine)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    prefetch_phase(0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    int st = 0;
    auto chunk = [&](int t, auto par_tag) {
        constexpr int par = decltype(par_tag)::value;
        using P0 = std::integral_constant<int, par>;
        using P1 = std::integral_constant<int, par ^ 1>;
        using H0 = std::integral_constant<int, 0>;
        using H1 = std::integral_constant<int, 1>;
        const int stn = st + 1 == S ? 0 : st + 1;
        if constexpr (NH == 2) {
            prefetch_phase(st, P0{}, H1{});
            __builtin_amdgcn_sched_barrier(0);
            mma_phase(P0{}, H0{});
            __builtin_amdgcn_sched_barrier(0);
        }
        // last phase of the chunk: chunk t + 1 must have landed (for every wave) and every wave must be done with slot st
        if (t + 1 < nsteps) {
            // steady state: chunks t + 2, t + 3 stay in flight; in the tail (nothing left to issue) simply drain
            if (t + S > nsteps) wait_vmcnt<0>();
            else if (xw) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * (Cfg::LPT + 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * Cfg::LPT) : "memory");
            __builtin_amdgcn_s_waitcnt(0xc07f);               // my fragment reads of slot st are done
#if VMV_XGLDS_ABLATE != 4
            __builtin_amdgcn_s_barrier();
#endif
            asm volatile("" ::: "memory");
            prefetch_phase(stn, P1{}, H0{});
            __builtin_amdgcn_sched_barrier(0);
        }
#if VMV_XGLDS_VARIANT == 1 || VMV_XGLDS_VARIANT == 3
        if (issued < nsteps) { issue(st); ++issued; }
        __builtin_amdgcn_sched_barrier(0);
#endif
        mma_phase(P0{}, std::integral_constant<int, NH - 1>{});
        __builtin_amdgcn_sched_barrier(0);
#if VMV_XGLDS_ABLATE == 2
        ++issued;
#elif VMV_XGLDS_VARIANT == 1 || VMV_XGLDS_VARIANT == 3
#else
        if (issued < nsteps) { issue(st); ++issued; }         // chunk t + S into the slot the barrier freed
#endif
        __builtin_amdgcn_s_waitcnt(0xc07f);
        st = stn;
    };
    for (int t = 0; t < nsteps; t += 2) {
        chunk(t, std::integral_constant<int, 0>{});
        chunk(t + 1, std::integral_constant<int, 1>{});
    }

    // ---- epilogue
    const int mbase = m0 + wave_m * 16 * WM + frow;
    if constexpr (SK) {           // raw partial sums -> this split's slab; the reduce pass owns bias / activation / residual / stores
        float* ws = p.workspace + (size_t)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                const int m = mbase + 16 * i, n = n0 + wave_n * 16 * WN + 16 * j + 4 * fgrp;
                if (m < p.M && n < p.N) *reinterpret_cast<f32x4_t*>(ws + (size_t)m * p.N + n) = acc[j][i];
            }
        return;
    }
    // (the launcher only sends GEMMs whose outputs can be staged: 16-bit, 16-byte aligned rows)
    // 16-bit outputs go through LDS (whole rows, 16-byte lanes, residual read the same way), half a tile (the two waves of
    // one wave row) at a time: the ring holds 128 rows of BN outputs
    constexpr bool GEGLU = (EPI & XE_GEGLU) != 0, LNF = (EPI & XE_LN) != 0;
    constexpr int BNO = GEGLU ? BN / 2 : BN;        // output columns of the tile (GEGLU: x | gate pairs -> one column)
    constexpr int row_bytes = BNO * 2 + 16;         // +16 B: spreads the 8-byte writes over the banks
    constexpr int U = BNO >> 3;                     // 16-byte units per tile row
    const int n0o = GEGLU ? n0 / 2 : n0, No = GEGLU ? p.N / 2 : p.N;
    uint16_t* outp = reinterpret_cast<uint16_t*>(p.out);
    const uint16_t* resp = reinterpret_cast<const uint16_t*>(p.residual);
    const float rs = p.res_scale != 0.f ? p.res_scale : 1.f;
    // Column vectors of the epilogue — bias plus, for conv1 of a ResBlock, the per-image embedding row (rowvec) — are staged
    // once per tile in LDS behind the staging area, one [BN] vector per row group the tile's rows span (<= XG_MAXG, launcher).
    // (Round 2 loaded them from global memory inside the (column tile, row tile) loops: exec-masked loads that the compiler
    //  follows with vmcnt(0) each — 40 dependent L2 round trips per wave — and whose hoisted 64-bit row pointers were the
    //  kernel's 10 spilled registers.)
    float* cvec = reinterpret_cast<float*>(smem + Cfg::HALF_ROWS * (BN * 2 + 16));
    float* csum = cvec + Cfg::XG_MAXG * BN;         // folded LayerNorm: the column sums of W' (vmv.h)
    const int rv_div = p.rowvec ? p.rowvec_div : (1 << 30);
    const int g0 = m0 / rv_div;
    const int m_last = (m0 + BM < p.M ? m0 + BM : p.M) - 1;
    const int ng = m_last / rv_div - g0 + 1;
    __syncthreads();                                // ring no longer read by anyone
    for (int idx = tid; idx < ng * BN; idx += Cfg::NT) {
        const int gq = idx / BN, n = n0 + (idx - gq * BN);
        float v = 0.f;
        if (n < p.N) {
            if (p.bias) v = p.bias[n];
            if (p.rowvec) v += p.rowvec[(size_t)(g0 + gq) * p.rowvec_ld + n];
        }
        cvec[idx] = v;
    }
    if constexpr (LNF) {
        for (int idx = tid; idx < BN; idx += Cfg::NT) csum[idx] = (n0 + idx) < p.N ? p.colsum[n0 + idx] : 0.f;
    }
    int gi[WM];                                     // row group (relative to g0) of this lane's row in each row tile
#pragma unroll
    for (int i = 0; i < WM; ++i) { const int m = mbase + 16 * i; gi[i] = ((m < p.M ? m : m_last) / rv_div - g0) * BN; }
    float ln_mean[WM], ln_rstd[WM];                 // folded LayerNorm: (mean, rstd) of this lane's four rows
    if constexpr (LNF) {
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int m = mbase + 16 * i < p.M ? mbase + 16 * i : m_last;
            const f32x2_t st2 = *reinterpret_cast<const f32x2_t*>(p.rowstat + 2 * (size_t)m);
            ln_mean[i] = st2.x; ln_rstd[i] = st2.y;
        }
    }
    auto finish = [&](const int j, const int i) -> f32x4_t {     // accumulator tile (j, i) after LN fold / column vector / activation
        const int nl = wave_n * 16 * WN + 16 * j + 4 * fgrp;     // column inside the tile
        f32x4_t v = acc[j][i];
        if constexpr (LNF) {
            const f32x4_t cs = *reinterpret_cast<const f32x4_t*>(csum + nl);
            v = (v - cs * ln_mean[i]) * ln_rstd[i];
        }
        v += *reinterpret_cast<const f32x4_t*>(cvec + gi[i] + nl);
        act_apply(v, p.act);
        return v;
    };
    // Residual rows run RD store-loop iterations AHEAD of their use (round 5), the first RD of a half tile requested BEFORE that half is
    // staged: instead of one dependent global-memory round trip per iteration (the temporal convolution block's last conv adds a
    // tensor written four launches earlier — out of L2 and, at the first level, out of the Infinity Cache: +28 us on a 95-us launch)
    // RD 16-byte loads per lane are in flight.  (All ITER at once would need 40 registers beside the 160 accumulators the other half's
    // waves still hold: 30 spilled.)  Buffer loads with out-of-range offsets for the tile tails.
    constexpr int ITER = Cfg::HALF_ROWS * U / Cfg::NT;
    constexpr int RD = EPI != 0 ? 1 : (ITER < 4 ? ITER : 4);      // (the folded-LayerNorm / GEGLU instantiations never carry a residual in the
                                                                   //  plans and have no registers to spare: one load ahead, as before)
    static_assert(ITER * Cfg::NT == Cfg::HALF_ROWS * U, "store loop trip count");
    const __amdgpu_buffer_rsrc_t res_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(resp), 0, resp ? SRD_RECORDS : 0u, SRD_FLAGS);
    auto res_request = [&](const int hh, const int it) -> u32x4_t {
        const int idx = tid + it * Cfg::NT;
        const int r = idx / U, u = idx - r * U;
        const int m = m0 + hh * Cfg::HALF_ROWS + r, n = n0o + u * 8;
        return __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, (m < p.M && n < No) ? (uint32_t)(m * p.ldr + n) * 2u : OOB, 0, 0);
    };
#pragma unroll 1
    for (int hh = 0; hh < BM / Cfg::HALF_ROWS; ++hh) {
        u32x4_t rr[RD];
        if (resp) {
#pragma unroll
            for (int it = 0; it < RD; ++it) rr[it] = res_request(hh, it);
        }
        __syncthreads();                            // column vectors written / the previous half's slab no longer read by anyone
        if ((wave_m >> 1) == hh) {
            if constexpr (GEGLU) {
#pragma unroll
                for (int jj = 0; jj < WN / 2; ++jj) {
#pragma unroll
                    for (int i = 0; i < WM; ++i) {
                        const f32x4_t x = finish(2 * jj, i), gt = finish(2 * jj + 1, i);
                        u32x2_t o;
                        o.x = pack_elem2(x.x * gelu_erf_f(gt.x), x.y * gelu_erf_f(gt.y));
                        o.y = pack_elem2(x.z * gelu_erf_f(gt.z), x.w * gelu_erf_f(gt.w));
                        *reinterpret_cast<u32x2_t*>(smem + ((wave_m & 1) * 16 * WM + 16 * i + frow) * row_bytes + (wave_n * 8 * WN + 16 * jj + 4 * fgrp) * 2) = o;
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < WN; ++j) {
#pragma unroll
                    for (int i = 0; i < WM; ++i) {
                        const f32x4_t v = finish(j, i);
                        u32x2_t o;
                        o.x = pack_elem2(v.x, v.y); o.y = pack_elem2(v.z, v.w);
                        *reinterpret_cast<u32x2_t*>(smem + ((wave_m & 1) * 16 * WM + 16 * i + frow) * row_bytes + (wave_n * 16 * WN + 16 * j + 4 * fgrp) * 2) = o;
                    }
                }
            }
        }
        __syncthreads();
        // Store-data discipline (see gemm_pglds.hip): the stored registers are a VALU-written copy, never the destination of an
        // LDS read, and the previous iteration's copy stays alive until this iteration's LDS read has returned.
        u32x4_t sd_prev = u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int idx = tid + it * Cfg::NT;
            const int r = idx / U, u = idx - r * U;
            const int m = m0 + hh * Cfg::HALF_ROWS + r, n = n0o + u * 8;
            if (m >= p.M || n >= No) continue;
            u32x4_t v = *reinterpret_cast<const u32x4_t*>(smem + r * row_bytes + u * 16);
            if (resp) {
                float a[8], b[8];
                unpack8(v, a); unpack8(rr[it % RD], b);
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] += rs * b[e];
                v = pack8(a);
                if (it + RD < ITER) rr[it % RD] = res_request(hh, it + RD);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            asm volatile("" ::"v"(sd_prev));
            u32x4_t sd;
            asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                         : "=&v"(sd.x), "=&v"(sd.y), "=&v"(sd.z), "=&v"(sd.w)
                         : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
            *reinterpret_cast<u32x4_t*>(outp + (size_t)m * p.ldo + n) = sd;
            sd_prev = sd;
        }
    }
}

// rows of the tile group that shares W slices in an XCD's L2 (kernel header): minimises the bytes 32 concurrent blocks pull in,
// gm x BM + 32 / gm x BN per K chunk, with 32 / gm <= the N tiles there are; 1 = the round-5 order.  VMV_XGLDS_GM forces it (A/B).
int xglds_group_m(int tiles_m, int tiles_n, int BM, int BN, int conc = 32) {
    static int env = -2;
    if (env == -2) { const char* e = getenv("VMV_XGLDS_GM"); env = e ? atoi(e) : -1; }
    if (env >= 1) return env;
    if (tiles_n < 2 || tiles_m < 2) return 1;
    int best = 1, best_cost = BM + conc * BN;          // conc = blocks an XCD runs at once (32 CUs x blocks per CU)
    for (int gm = 2; gm <= conc; gm *= 2) {
        const int gn = (conc + gm - 1) / gm;
        if (gn > tiles_n || gm > tiles_m) continue;
        const int cost = gm * BM + gn * BN;
        if (cost < best_cost) { best = gm; best_cost = cost; }
    }
    return best;
}

int xglds_tapmajor() {        // VMV_XGLDS_TAPMAJOR (A/B): 1 = tap-interleaved K walk of the convolutions (kernel header), 0 = segment-major
    static int v = -1;
    if (v < 0) { const char* e = getenv("VMV_XGLDS_TAPMAJOR"); v = e ? atoi(e) : 1; }
    return v;
}

template <int NH, int WH, int EPI = 0, int WNV = 2>
int launch_xglds(const VmvGemmParams& p, int total_steps, hipStream_t st) {
    using Cfg = WgCfg<NH, WH, WNV>;
    const int tiles_m = (p.M + Cfg::BM - 1) / Cfg::BM;
    const int tiles_n = (p.N + Cfg::BN - 1) / Cfg::BN;
    const int gm = xglds_group_m(tiles_m, tiles_n, Cfg::BM, Cfg::BN, 32 * Cfg::BPC);
    int nsteps = 0;                                          // chunks of WBK = 32 (total_steps counts the other kernels' 64)
    for (int i = 0; i < p.nseg; ++i) nsteps += (p.seg[i].k + WBK - 1) / WBK;
    (void)total_steps;
    if (nsteps < Cfg::STAGES || (nsteps & 1)) return VMV_GLDS_UNSUPPORTED;
    if (p.rowvec && (Cfg::BM - 1) / p.rowvec_div + 2 > Cfg::XG_MAXG) return VMV_GLDS_UNSUPPORTED;      // column vectors staged per row group
    if (p.ksplit > 1) {
        if constexpr (EPI != 0 || WNV != 2) return VMV_GLDS_UNSUPPORTED;
        else {
            // even chunk counts per split (the loop body is two chunks), every split >= STAGES chunks, no empty split
            int sps = (nsteps + p.ksplit - 1) / p.ksplit;
            sps += sps & 1;
            const int last = nsteps - (p.ksplit - 1) * sps;
            if (sps < Cfg::STAGES || last < Cfg::STAGES || (last & 1)) return VMV_GLDS_UNSUPPORTED;
            static std::atomic<unsigned long long> attr_sk{0};
            if (const int rc_attr = vmv_lds_attr_once(attr_sk, reinterpret_cast<const void*>(&gemm_xglds_kernel<NH, WH, 0, true>), Cfg::LDS_BYTES)) return rc_attr;
            VMV_LAUNCH((gemm_xglds_kernel<NH, WH, 0, true>), dim3(tiles_m * tiles_n, p.ksplit), dim3(Cfg::NT), Cfg::LDS_BYTES, st, p, tiles_m, tiles_n,
                               sps, nsteps, gm, 0);
            return vmv_launch_status();
        }
    }
    static std::atomic<unsigned long long> attr_set{0};
    if (const int rc_attr = vmv_lds_attr_once(attr_set, reinterpret_cast<const void*>(&gemm_xglds_kernel<NH, WH, EPI, false, WNV>), Cfg::LDS_BYTES)) return rc_attr;
    VMV_LAUNCH((gemm_xglds_kernel<NH, WH, EPI, false, WNV>), dim3(tiles_m * tiles_n), dim3(Cfg::NT), Cfg::LDS_BYTES, st, p, tiles_m, tiles_n, nsteps, nsteps, gm, xglds_tapmajor());
    return vmv_launch_status();
}

}  // namespace

// 1 if the wide-tile kernel has an instantiation for *p's epilogue on `tile` (host logic; used by the policy in gemm.hip)
int vmv_gemm_xglds_epi_ok(const VmvGemmParams& p, int tile) {
    const bool geglu = p.epilogue == VMV_EPI_GEGLU, lnf = p.rowstat != nullptr;
    if (!geglu && !lnf) return 1;
    if (tile != VMV_TILE_X256x256 && tile != VMV_TILE_Y256x128) return 0;      // the fused epilogues exist for 64 x 128 wave tiles only
    if (lnf && !p.colsum) return 0;
    if (geglu && ((p.N & 31) || p.residual || p.act != VMV_ACT_NONE)) return 0;      // whole x | gate pairs; no residual (as the other kernels)
    return 1;
}

// Called by vmv_gemm (gemm.hip) after argument validation.
int vmv_gemm_xglds_launch(const VmvGemmParams& p, int total_steps, int tile, hipStream_t st) {
    if (vmv_gemm_ln_inline(p) || !vmv_gemm_xglds_epi_ok(p, tile)) return VMV_GLDS_UNSUPPORTED;
    const bool geglu = p.epilogue == VMV_EPI_GEGLU, lnf = p.rowstat != nullptr;
    if (p.ksplit > 1 && (geglu || lnf || !p.workspace)) return VMV_GLDS_UNSUPPORTED;      // split-K: plain epilogue only (the reduce pass runs it)
    long maxrows = p.M;
    if (p.OH > 0) { const long src_rows = (long)(p.M / (p.OH * p.OW) + 1) * p.IH * p.IW; if (src_rows > maxrows) maxrows = src_rows; }
    for (int i = 0; i < p.nseg; ++i)
        if (maxrows * (long)p.seg[i].ld * 2 >= (1L << 31) - 65536) return VMV_GLDS_UNSUPPORTED;
    if ((long)(p.N + 320) * p.ktot * 2 >= (1L << 31) - 65536) return VMV_GLDS_UNSUPPORTED;
    if (p.OH > 0 && (p.OH >= 32768 || p.OW >= 32768)) return VMV_GLDS_UNSUPPORTED;      // (oy, ox) packed in one register
    if (p.residual && (long)p.M * p.ldr * 2 >= (1L << 31) - 65536) return VMV_GLDS_UNSUPPORTED;      // residual rows by 32-bit buffer offsets
    const int No = geglu ? p.N / 2 : p.N;
    if (p.ksplit <= 1 && (p.out_fp32 || (p.ldo & 7) || (No & 7) || !vmv_aligned16(p.out) ||
                          (p.residual && ((p.ldr & 7) || !vmv_aligned16(p.residual))))) return VMV_GLDS_UNSUPPORTED;   // staged epilogue only
    if (tile == VMV_TILE_Y256x128) {          // 4-wave blocks, two per CU (WgCfg WNV = 4): measured and rejected, experiment builds only
#if defined(VMV_EXPERIMENTS)
        if (geglu && lnf) return launch_xglds<2, 4, XE_GEGLU | XE_LN, 4>(p, total_steps, st);
        if (geglu) return launch_xglds<2, 4, XE_GEGLU, 4>(p, total_steps, st);
        if (lnf) return launch_xglds<2, 4, XE_LN, 4>(p, total_steps, st);
        return launch_xglds<2, 4, 0, 4>(p, total_steps, st);
#else
        return VMV_GLDS_UNSUPPORTED;
#endif
    }
    if (geglu || lnf) {
        if (geglu && lnf) return launch_xglds<2, 4, XE_GEGLU | XE_LN>(p, total_steps, st);
        if (geglu) return launch_xglds<2, 4, XE_GEGLU>(p, total_steps, st);
        return launch_xglds<2, 4, XE_LN>(p, total_steps, st);
    }
    if (tile == VMV_TILE_X256x320) return launch_xglds<2, 5>(p, total_steps, st);
    if (tile == VMV_TILE_X256x256) return launch_xglds<2, 4>(p, total_steps, st);
    if (tile == VMV_TILE_X256x128) return launch_xglds<1, 4>(p, total_steps, st);
    if (tile == VMV_TILE_X512x128) return launch_xglds<2, 4, 0, 1>(p, total_steps, st);
    return VMV_EINVAL;
}

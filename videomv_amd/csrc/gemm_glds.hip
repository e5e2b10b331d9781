// gemm_glds.hip — the LDS-DMA variants of the bf16 MFMA implicit GEMM (same contract as gemm.hip / vmv.h).
//
//   * wave tile 64 x {64,80} of v_mfma_f32_16x16x32_bf16 (transposed product, see gemm.hip), BK = 64, and two block shapes:
//       - 8 waves (4 along M x 2 along N), tile 256 x {128,160}, 3-stage ring (144/156 KB): ONE block per CU — the
//         long-K workhorse (conv3x3 / temporal conv / FF-down): least LDS-fill traffic per flop, loads two chunks ahead;
//       - 4 waves (2 x 2), tile 128 x {128,160}, 2-stage ring (64/72 KB): TWO blocks per CU — for the short-K linears
//         (K = C = 320/640: 5-10 chunks per tile), where a tile's fill + epilogue (GEGLU: ~as many VALU cycles as the
//         tile has MFMA cycles) is as long as its main loop; two co-resident blocks run one's epilogue / prologue
//         under the other's MFMAs.
//   * operands go global -> LDS directly (LDS-DMA, global_load_lds_dwordx4): no staging VGPRs, no ds_write pass.
//     Zero fill (conv padding, M/N/K tails) is done by pointing the lane at a zero page instead of predicating.
//     The LDS image is lane-linear per wave instruction (8 rows x 128 B), so the XOR swizzle that makes the
//     ds_read_b128 fragment reads conflict-free is applied to the per-lane SOURCE address.
//   * 3-stage LDS ring (3 x 48/52 KB), loads run TWO chunks ahead of the MFMAs with counted s_waitcnt vmcnt(N)
//     and one raw s_barrier per chunk:   wait(chunk t landed) -> barrier -> issue(chunk t+2) -> MFMA(chunk t).
#include "gemm_glds_common.h"
#include <cstdlib>
#include <type_traits>

using namespace vmvg;

namespace {

template <int WMW, int WN, int STAGES, int ablate, bool PP = false>
__global__ __launch_bounds__(128 * WMW, WMW == 2 ? 2 : 1) void gemm_glds_kernel(const VmvGemmParams p, const int tiles_m, const int tiles_n,
                                                              const int total_steps, const int steps_per_split, const int gm, const int tapmajor) {
    VMV_KERNEL_ENTER();
    // ablate (experiments only, VMV_GEMM_ABLATE): 1 = skip the MFMAs + fragment reads, 2 = skip the LDS-DMA loads
    using Cfg = GlCfg<WMW, WN, STAGES>;
    constexpr int WM = 4;                         // 16-row MFMA tiles per wave along M (wave tile = 64 rows)
    constexpr int BN = Cfg::BN;
    constexpr int GL_BM = Cfg::BM;
    constexpr int GL_STAGES = STAGES;
    constexpr int NW = Cfg::NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;

    // ---- XCD-aware tile mapping (bijective)
    const int nblk = tiles_m * tiles_n;
    int logical;
    {
        const int bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // grouped order (round 6; gemm_xglds.hip): gm row tiles x 32 / gm column tiles run together on an XCD, so its L2 serves each W
    // slice to gm row tiles instead of one (gm = 1: the N tiles of one row tile adjacent, as before)
    int tile_m, tile_n;
    if (gm > 1) {
        const int gsz = gm * tiles_n, g = logical / gsz, first = g * gm;
        const int gmh = tiles_m - first < gm ? tiles_m - first : gm;
        const int rem = logical - g * gsz;
        tile_n = rem / gmh; tile_m = first + (rem - tile_n * gmh);
    } else {
        tile_n = logical % tiles_n; tile_m = logical / tiles_n;
    }
    const int m0 = tile_m * GL_BM, n0 = tile_n * BN;
    const int split = blockIdx.y;
    const int step_begin = split * steps_per_split;
    const int step_end = min(total_steps, step_begin + steps_per_split);
    const int nsteps = step_end - step_begin;

    // ---- load assignment.  Wave instruction covers 8 rows x 128 B; lane -> (row-in-group = lane>>3, physical slot =
    //      lane&7).  Row r stores logical slot s at physical slot s ^ ((r>>1)&7); for every row this lane touches
    //      (r = 8*g + (lane>>3), g = NW*i + wave [- NW], NW even) that XOR term is ((wave&1)*4 + (lane>>4)) & 7.
    const int lrow = lane >> 3;
    const int lsw = (lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);    // logical 16-B slot this lane fetches

    uint32_t wvo[Cfg::NWI];                       // per-lane byte offset of its weight row (+ its k-slot) or OOB
    int wgrp[Cfg::NWI];
#pragma unroll
    for (int j = 0; j < Cfg::NWI; ++j) {
        int g = j * NW + wave;
        if (g >= BN / 8) g -= NW;                // duplicate an earlier row group: keeps the per-wave load count uniform
        wgrp[j] = g;
        const int n = n0 + g * 8 + lrow;
        // (grouped weights, vmv.h: the rows of this tile multiply the weight matrix of group m0 / wgroup_rows)
        const uint32_t wg = p.wgroup_rows > 0 ? (uint32_t)((long)(m0 / p.wgroup_rows) * p.wgroup_stride) * 2u : 0u;
        wvo[j] = (n < p.N) ? (uint32_t)(n * p.ktot + lsw * 8) * 2u + wg : OOB;
    }
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, SRD_RECORDS, SRD_FLAGS);

    // ---- K-walk state (runs two chunks ahead of the MFMAs).  The walk is RUN-wise (round 6, as gemm_xglds.hip): consecutive segments
    //      that differ only in their tap (the 9 of a 3 x 3 convolution, the 3 of a temporal one) form a run and are walked chunk-major —
    //      the 64 channels of chunk c for every tap, then chunk c + 1 — so the taps of a chunk re-read the tile's rows out of the L2
    //      instead of the fabric; W keeps its (tap, channel) K order and is read at column tap * C + 64 c.  Everything else (linear,
    //      strided / up-sampling taps) is a run of one.  Per lane and A row: one byte offset (the row's tap-(0, 0) position; for a run
    //      of one the tap's own position) and a validity bit per tap, computed once per run from m (no per-row state is kept).
    int s = 0, koff = 0;
    int run_len = 1, run_t = 0, run_nch = 0, run_c = 0;
    auto seg_same = [&](const VmvGemmSeg& a, const VmvGemmSeg& b) { return a.src == b.src && a.ld == b.ld && a.k == b.k && a.mode == b.mode; };
    auto detect_run = [&]() {               // run_len / run_nch of the run that starts at segment s
        const VmvGemmSeg& s0 = p.seg[s];
        run_len = 1;
        run_nch = (s0.k + BK - 1) / BK;
        const bool tappable = (s0.mode == VMV_SEG_SPATIAL && p.stride == 1 && p.ups == 0) || s0.mode == VMV_SEG_TEMPORAL;
        if (tapmajor && tappable)
            while (s + run_len < p.nseg && run_len < 15 && seg_same(p.seg[s + run_len], s0)) ++run_len;
    };
    static_assert(Cfg::NAI == 4, "four 16-bit tap masks in two registers");
    uint32_t avo[Cfg::NAI];                       // per-lane byte offset of its gathered source row (+ k-slot)
    uint32_t rmask[2] = {0u, 0u};                 // tap-validity bits of the lane's four rows, 16 per row
    auto setup_rows = [&]() {
        const VmvGemmSeg& s0 = p.seg[s];
        const int mode = s0.mode, ld = s0.ld;
        const bool single = run_len == 1;
        rmask[0] = 0u; rmask[1] = 0u;
#pragma unroll
        for (int i = 0; i < Cfg::NAI; ++i) {
            int m = m0 + (i * NW + wave) * 8 + lrow;
            asm volatile("" : "+v"(m));             // keeps the divisions below out of the K loop's live registers
            const bool inm = m < p.M;
            int nb = 0, oy = 0, ox = 0;             // spatial: image base row n IH IW, output (y, x); temporal: nb = frame index
            if (mode == VMV_SEG_SPATIAL) {
                const int hw = p.OH * p.OW;
                const int n = m / hw, rem = m - n * hw;
                oy = rem / p.OW; ox = rem - oy * p.OW;
                nb = n * p.IH * p.IW;
            } else if (mode == VMV_SEG_TEMPORAL) {
                nb = (m / p.P) % p.F;
            }
            uint32_t mask = 0;
            int base;
            if (mode == VMV_SEG_LINEAR) {
                base = m * ld; mask = inm ? 1u : 0u;
            } else if (mode == VMV_SEG_SPATIAL) {
                if (single) {
                    const int iy = oy * p.stride + s0.d0, ix = ox * p.stride + s0.d1;
                    const int VH = p.IH << p.ups, VW = p.IW << p.ups;
                    const bool ok = inm && iy >= 0 && iy < VH && ix >= 0 && ix < VW;
                    base = ok ? (nb + (iy >> p.ups) * p.IW + (ix >> p.ups)) * ld : 0;
                    mask = ok ? 1u : 0u;
                } else {
                    base = (nb + oy * p.IW + ox) * ld;
                    for (int t = 0; t < run_len; ++t) {
                        const int iy = oy + p.seg[s + t].d0, ix = ox + p.seg[s + t].d1;
                        if (inm && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW) mask |= 1u << t;
                    }
                }
            } else {
                if (single) {
                    const int f = nb + s0.d0;
                    const bool ok = inm && f >= 0 && f < p.F;
                    base = ok ? (m + s0.d0 * p.P) * ld : 0;
                    mask = ok ? 1u : 0u;
                } else {
                    base = m * ld;
                    for (int t = 0; t < run_len; ++t) {
                        const int f = nb + p.seg[s + t].d0;
                        if (inm && f >= 0 && f < p.F) mask |= 1u << t;
                    }
                }
            }
            avo[i] = (uint32_t)(base + lsw * 8) * 2u;
            rmask[i >> 1] |= mask << (16 * (i & 1));
        }
    };
    {       // fast-forward to this split's first chunk (split-K: chunks are counted in walk order)
        int skip = step_begin;
        while (s < p.nseg) {
            detect_run();
            const int total = run_nch * run_len;
            if (skip < total) { run_c = skip / run_len; run_t = skip - run_c * run_len; break; }
            skip -= total; koff += run_len * p.seg[s].k; s += run_len;
        }
    }
    if (s < p.nseg && nsteps > 0) setup_rows();
    auto advance = [&](const int segk) {    // the K walk moves one chunk on
        if (++run_t == run_len) {
            run_t = 0;
            if (++run_c == run_nch) {
                koff += run_len * segk; s += run_len; run_c = 0;
                if (s < p.nseg) { detect_run(); setup_rows(); }
            }
        }
    };
    // what one chunk's loads need besides the lane offsets: the segment's descriptor, the scalar offsets, the tap's distance from the
    // run's base position (elements x 2, wave-uniform; may be negative: it is added to the 32-bit lane offset, a valid tap's sum is a
    // non-negative offset inside the tensor) and the k-tail flag
    struct ChunkArgs { __amdgpu_buffer_rsrc_t a_rsrc; uint32_t a_so, w_so, d2; bool kvalid; int tapbit; int segk; };
    auto chunk_args = [&](const bool more) -> ChunkArgs {
        const VmvGemmSeg& sg = p.seg[more ? s + run_t : 0];
        ChunkArgs c;
        c.a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(sg.src), 0, SRD_RECORDS, SRD_FLAGS);
        const int kc = run_c * BK;
        c.kvalid = (kc + BK) <= sg.k || (kc + lsw * 8) < sg.k;      // k tail of a segment: zero fill
        int delta = 0;
        if (run_len > 1) delta = sg.mode == VMV_SEG_SPATIAL ? (sg.d0 * p.IW + sg.d1) * sg.ld : sg.d0 * p.P * sg.ld;
        c.a_so = (uint32_t)kc * 2u; c.w_so = (uint32_t)(koff + run_t * sg.k + kc) * 2u; c.d2 = (uint32_t)(delta * 2);
        c.tapbit = run_t; c.segk = sg.k;
        return c;
    };
    auto a_off = [&](const ChunkArgs& c, const int i) -> uint32_t {
        return (c.kvalid && ((rmask[i >> 1] >> (16 * (i & 1) + c.tapbit)) & 1u)) ? avo[i] + c.d2 : OOB;
    };

    auto issue = [&](int stage) {           // LDS-DMA one chunk into ring slot `stage`, then advance the walk
        const ChunkArgs c = chunk_args(true);
        unsigned char* abase = smem + stage * Cfg::STAGE_BYTES + wave * 1024;
        unsigned char* wbase = smem + stage * Cfg::STAGE_BYTES + Cfg::A_BYTES;
#pragma unroll
        for (int i = 0; i < Cfg::NAI; ++i) VMV_BLDS16(c.a_rsrc, abase + i * (NW * 1024), a_off(c, i), c.a_so);
#pragma unroll
        for (int j = 0; j < Cfg::NWI; ++j) VMV_BLDS16(w_rsrc, wbase + wgrp[j] * 1024, c.kvalid ? wvo[j] : OOB, c.w_so);
        advance(c.segk);
    };

    f32x4_t acc[WN][WM];
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int i = 0; i < WM; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15;
    const int fgrp = lane >> 4;
    const int fswz = (frow >> 1) & 7;

    // ---- software-pipelined main loop.  Fragment registers are double-buffered per k-step (kk = 0/1 of a 64-chunk):
    //   [ds_read kk1(t)] [MFMA kk0(t)] | chunk t+1 landed? -> lgkmcnt(0) -> s_barrier -> LDS-DMA chunk t+3 into the slot
    //   just freed | [ds_read kk0(t+1)] [MFMA kk1(t)]
    // so every MFMA batch runs under the LDS reads of the next one (the lock-step "all waves read, then all waves
    // multiply" phases of a plain loop leave the matrix pipe idle during the LDS burst), and the barrier of chunk t+1
    // sits between two MFMA batches that need no LDS.
    auto read_frags = [&](int slot_idx, int kk, elem8_t (&af)[WM], elem8_t (&wf)[WN]) {
        const u32x4_t* a = reinterpret_cast<const u32x4_t*>(smem + slot_idx * Cfg::STAGE_BYTES) + (wave_m * 64 + frow) * 8;
        const u32x4_t* w = reinterpret_cast<const u32x4_t*>(smem + slot_idx * Cfg::STAGE_BYTES + Cfg::A_BYTES) +
                           (wave_n * 16 * WN + frow) * 8;
        const int slot = (kk * 4 + fgrp) ^ fswz;
#pragma unroll
        for (int i = 0; i < WM; ++i) af[i] = __builtin_bit_cast(elem8_t, a[i * 16 * 8 + slot]);
#pragma unroll
        for (int j = 0; j < WN; ++j) wf[j] = __builtin_bit_cast(elem8_t, w[j * 16 * 8 + slot]);
    };
    auto mma = [&](const elem8_t (&af)[WM], const elem8_t (&wf)[WN]) {
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int i = 0; i < WM; ++i)
                acc[j][i] = VMV_MFMA16(wf[j], af[i], acc[j][i], 0, 0, 0);
    };

    elem8_t a0[WM], w0[WN], a1[WM], w1[WN];
    if constexpr (PP) {
        // ---- ping-pong schedule (8 waves; waves w and w+4 share a SIMD).  Every chunk has a LOAD phase (fragment reads of
        //      chunk t + LDS-DMA issue of chunk t+2) and a MATRIX phase (its 40 MFMAs), separated by block barriers; the
        //      waves 4-7 run one phase behind the waves 0-3, so on every SIMD one wave multiplies while its partner
        //      reads / issues.  Ring invariants: chunk t+1 is waited for by everyone before the barrier that opens the
        //      first group's LOAD(t+1); the slot of chunk t-1 is refilled only after both groups' LOAD(t-1).
        static_assert(!PP || (WMW == 4 && STAGES == 3), "ping-pong needs the 8-wave 3-stage block");
        const int grp = wave >> 2;
        int issued = 0;
        const int pro = nsteps < 2 ? nsteps : 2;
        for (int i = 0; i < pro; ++i) { issue(i); ++issued; }
        if (nsteps > 0) {
            if (pro == 2) wait_vmcnt<Cfg::LPT>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
            if (grp == 1) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            int st = 0;
            // ablate == 4: block 0 stamps s_memtime of its waves 0 and 4 at the phase edges of chunk 8 into p.workspace
            unsigned long long* stamps = reinterpret_cast<unsigned long long*>(p.workspace);
            auto stamp = [&](int t, int k) {
                if constexpr (ablate == 4) {
                    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && (wave & 3) == 0 && t == 8 && stamps)
                        stamps[grp * 8 + k] = __builtin_readcyclecounter();
                }
            };
            for (int t = 0; t < nsteps; ++t) {
                // LOAD(t): the chunk's 2*(WM+WN) fragment reads INTERLEAVED with the LPT LDS-DMA pieces of chunk t+2, so that the
                // LDS read port and the address/TA path work at the same time (issued back to back as two bursts they
                // serialise: measured 430 + 560 cycles per wave against 740 cycles of MFMAs in the partner's phase).
                stamp(t, 0);
                int s2 = st + 2; if (s2 >= 3) s2 -= 3;
                const bool more = issued < nsteps;
                {
                    const ChunkArgs c = chunk_args(more);
                    unsigned char* abase = smem + s2 * Cfg::STAGE_BYTES + wave * 1024;
                    unsigned char* wbase = smem + s2 * Cfg::STAGE_BYTES + Cfg::A_BYTES;
                    const u32x4_t* fa = reinterpret_cast<const u32x4_t*>(smem + st * Cfg::STAGE_BYTES) + (wave_m * 64 + frow) * 8;
                    const u32x4_t* fw = reinterpret_cast<const u32x4_t*>(smem + st * Cfg::STAGE_BYTES + Cfg::A_BYTES) +
                                        (wave_n * 16 * WN + frow) * 8;
                    const int slot0 = fgrp ^ fswz, slot1 = (4 + fgrp) ^ fswz;
                    constexpr int NRD = 2 * (WM + WN);
                    auto read_one = [&](int r) {               // r: a0[0..WM), w0[0..WN), a1[..], w1[..]
                        if (r < WM) a0[r] = __builtin_bit_cast(elem8_t, fa[r * 16 * 8 + slot0]);
                        else if (r < WM + WN) w0[r - WM] = __builtin_bit_cast(elem8_t, fw[(r - WM) * 16 * 8 + slot0]);
                        else if (r < 2 * WM + WN) a1[r - WM - WN] = __builtin_bit_cast(elem8_t, fa[(r - WM - WN) * 16 * 8 + slot1]);
                        else w1[r - 2 * WM - WN] = __builtin_bit_cast(elem8_t, fw[(r - 2 * WM - WN) * 16 * 8 + slot1]);
                    };
                    int rd = 0;
#pragma unroll
                    for (int k = 0; k < Cfg::LPT; ++k) {
                        if (more && ablate != 2) {
                            if (k < Cfg::NAI) VMV_BLDS16(c.a_rsrc, abase + k * (NW * 1024), a_off(c, k < Cfg::NAI ? k : 0), c.a_so);
                            else VMV_BLDS16(w_rsrc, wbase + wgrp[k - Cfg::NAI] * 1024, c.kvalid ? wvo[k - Cfg::NAI] : OOB, c.w_so);
                        }
                        const int upto = (NRD * (k + 1)) / Cfg::LPT;
                        if constexpr (ablate != 1) {
#pragma unroll
                            for (int r = 0; r < NRD; ++r) if (r >= rd && r < upto) read_one(r);
                        }
                        rd = upto;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    stamp(t, 1);
                    if (more) { advance(c.segk); ++issued; }
                }
                stamp(t, 2);
                if (grp == 1) { if (more) wait_vmcnt<Cfg::LPT>(); else wait_vmcnt<0>(); }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                stamp(t, 3);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                stamp(t, 4);
                // MATRIX(t)
                if constexpr (ablate != 1) { mma(a0, w0); mma(a1, w1); }
                __builtin_amdgcn_sched_barrier(0);
                stamp(t, 5);
                if (grp == 0) { if (more) wait_vmcnt<Cfg::LPT>(); else wait_vmcnt<0>(); }
                stamp(t, 6);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                stamp(t, 7);
                st = st + 1 == 3 ? 0 : st + 1;
            }
            if (grp == 0) __builtin_amdgcn_s_barrier();
        }
    } else {
        int issued = 0;
        const int pro = nsteps < GL_STAGES ? nsteps : GL_STAGES;
        for (int i = 0; i < pro; ++i) { issue(i); ++issued; }
        if (ablate == 2 || ablate >= 5) issued = nsteps;       // 5: MFMAs only; 6: MFMAs + fragment reads (no barriers, no DMA)
        if (nsteps > 0) {
            if (pro == 3) wait_vmcnt<(GL_STAGES == 3 ? 2 : 0) * Cfg::LPT>(); else if (pro == 2) wait_vmcnt<Cfg::LPT>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (ablate != 1 && ablate != 5) read_frags(0, 0, a0, w0);
            else if (ablate == 5) {
#pragma unroll
                for (int i = 0; i < WM; ++i) { a0[i] = elem8_t{}; a1[i] = elem8_t{}; }
#pragma unroll
                for (int j = 0; j < WN; ++j) { w0[j] = elem8_t{}; w1[j] = elem8_t{}; }
            }
        }
        int st = 0;                                   // ring slot of chunk t
        for (int t = 0; t + 1 < nsteps; ++t) {        // (the last chunk is peeled below: no control-flow merge in here)
            if constexpr (ablate != 1) {
                if constexpr (ablate != 5) read_frags(st, 1, a1, w1);
                __builtin_amdgcn_sched_barrier(0);
                mma(a0, w0);
            }
            int stn = st + 1; if (stn == GL_STAGES) stn = 0;
            // chunk t+1 landed (mine): chunks up to t+STAGES-1 are in flight, so a 3-stage ring may leave one outstanding
            if (GL_STAGES == 3 && t + 2 < nsteps) wait_vmcnt<(GL_STAGES == 3 ? 1 : 0) * Cfg::LPT>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0): my reads of slot st are done (builtin: the compiler's
                                                      // wait-count pass sees it and adds no second drain before the MFMAs)
            if constexpr (ablate < 5) __builtin_amdgcn_s_barrier();             // ... for every wave: slot st is free, chunk t+1 is visible
            asm volatile("" ::: "memory");
            if constexpr (ablate != 1) {
                if constexpr (ablate != 5) read_frags(stn, 0, a0, w0);
                __builtin_amdgcn_sched_barrier(0);
                mma(a1, w1);
                __builtin_amdgcn_sched_barrier(0);
            }
            // address arithmetic, scalar loads of the segment table and the LDS-DMA issue run in the shadow of the
            // MFMA batch just issued (chunk t+3 -> the slot freed by the barrier above)
            if (issued < nsteps) { issue(st); ++issued; }
            __builtin_amdgcn_s_waitcnt(0xc07f);       // retire issue()'s scalar loads here (and the long-finished a0/w0
                                                      // reads) so the next iteration's first MFMA batch needs no drain
            st = stn;
        }
        if (nsteps > 0) {
            if constexpr (ablate != 1) {
                if constexpr (ablate != 5) read_frags(st, 1, a1, w1);
                mma(a0, w0);
                mma(a1, w1);
            }
        }
    }

    // ---- epilogue (identical to gemm.hip)
    const int mbase = m0 + wave_m * 64 + frow;
    const int nbase = n0 + wave_n * 16 * WN + 4 * fgrp;
    if (p.ksplit > 1) {
        float* ws = p.workspace + (size_t)split * p.M * p.N;
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                const int m = mbase + 16 * i, n = nbase + 16 * j;
                if (m < p.M && n < p.N) *reinterpret_cast<f32x4_t*>(ws + (size_t)m * p.N + n) = acc[j][i];
            }
        return;
    }
    // ---- bf16 outputs go through LDS: the MFMA layout gives every lane 4 channels of one row (8-byte pieces, a
    //      128-B line is touched by 4 different store instructions); staging the tile lets the block write whole
    //      rows with 16-byte lanes and read the residual the same way.  For the short-K linears (5 chunks per tile) the
    //      epilogue is as long as the main loop, so this matters.
    const bool geglu = p.epilogue == VMV_EPI_GEGLU;
    const int out_w = geglu ? BN / 2 : BN;                       // output columns of this tile
    const int n_out0 = geglu ? n0 / 2 : n0;
    const int N_out = geglu ? p.N / 2 : p.N;
    const bool staged = !p.out_fp32 && (p.ldo & 7) == 0 && (N_out & 7) == 0 && vmv_ptr_aligned16(p.out) &&
                        (!p.residual || ((p.ldr & 7) == 0 && vmv_ptr_aligned16(p.residual)));
    if (!staged) {
        if (geglu) {
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
    __syncthreads();                                             // ring no longer read by anyone
    const int row_bytes = out_w * 2 + 16;                        // +16 B: spreads the 8-byte writes over the banks
    {
        const int trow0 = wave_m * 64 + frow;
        const int tcol0 = wave_n * 16 * WN + 4 * fgrp;
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            if (geglu && (j & 1)) continue;
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                const int m = mbase + 16 * i, n = nbase + 16 * j;
                f32x4_t v = acc[j][i];
                int tc = tcol0 + 16 * j;
                if (n < p.N) {
                    if (p.bias) v += *reinterpret_cast<const f32x4_t*>(p.bias + n);
                    int no = n;
                    if (geglu) {
                        if constexpr ((WN & 1) == 0) {
                            f32x4_t g = acc[(j + 1) % WN][i];
                            if (p.bias) g += *reinterpret_cast<const f32x4_t*>(p.bias + n + 16);
                            if constexpr (ablate == 3) { v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w; }    // (experiment: no GELU)
                            else { v.x *= gelu_erf_f(g.x); v.y *= gelu_erf_f(g.y); v.z *= gelu_erf_f(g.z); v.w *= gelu_erf_f(g.w); }
                        }
                        no = (n >> 5) * 16 + (n & 15);
                        tc = (tc >> 5) * 16 + (tc & 15);
                    }
                    if (p.rowvec && m < p.M)
                        v += *reinterpret_cast<const f32x4_t*>(p.rowvec + (size_t)(m / p.rowvec_div) * p.rowvec_ld + no);
                    act_apply(v, p.act);
                } else if (geglu) {
                    tc = (tc >> 5) * 16 + (tc & 15);
                }
                u32x2_t o;
                o.x = pack_elem2(v.x, v.y); o.y = pack_elem2(v.z, v.w);
                *reinterpret_cast<u32x2_t*>(smem + (trow0 + 16 * i) * row_bytes + tc * 2) = o;
            }
        }
    }
    __syncthreads();
    {
        const int U = out_w >> 3;                                // 16-byte units per tile row
        uint16_t* outp = reinterpret_cast<uint16_t*>(p.out);
        const uint16_t* resp = reinterpret_cast<const uint16_t*>(p.residual);
        const float rs = p.res_scale != 0.f ? p.res_scale : 1.f;
        // Store-data discipline (see gemm_pglds.hip): the stored registers are a VALU-written copy, never the destination
        // of an LDS read, and the previous iteration's copy stays alive until this iteration's LDS read has returned — an
        // LDS read returning into a pending store's data registers corrupts the store when the store path is backed up.
        u32x4_t sd_prev = u32x4_t{0u, 0u, 0u, 0u};
        for (int idx = tid; idx < GL_BM * U; idx += Cfg::NT) {
            const int r = idx / U, u = idx - r * U;
            const int m = m0 + r, n = n_out0 + u * 8;
            if (m >= p.M || n >= N_out) continue;
            u32x4_t v = *reinterpret_cast<const u32x4_t*>(smem + r * row_bytes + u * 16);
            if (resp) {
                const u32x4_t rr = *reinterpret_cast<const u32x4_t*>(resp + (size_t)m * p.ldr + n);
                float a[8], b[8];
                unpack8(v, a); unpack8(rr, b);
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] += rs * b[e];
                v = pack8(a);
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

// Run-wise K walk of the convolutions (kernel): on for M >= 16384 rows, off below; VMV_GLDS_TAPMAJOR = 0 / 1 forces it (A/B).
// Measured on one box: the walk cuts this kernel's convolution fetches 5.2 x -> 3.8 x of the algorithmic bytes at the UNet's third /
// fourth level (what is left is W, streamed once into each of the 8 L2s: 240 tiles = one round), but those convolutions (M = 7680 /
// 1920) run 1-4 % slower (L3 tconv 314 -> 303 TFLOP/s, conv L2 1018 -> 1008) and the step +0.25 ms (48.05 / 48.20 vs 47.87 / 47.89;
// profiles/r6_tap2_*.log, r6b_gemm_traffic_by_kernel.tsv): there the fetches it saves come out of the Infinity Cache and were not
// the bound.  Where A is the whole traffic — the VAE's 128-channel convolutions over 1.6-3.1 M rows (W = 295 KB) — it pays: encoder
// first level 570 -> 672 / 503 -> 585 TFLOP/s, 48-view encode 22.4 -> 20.6 ms, decode 7.7 -> 7.4 (profiles/r6_lgm_step_bench*.log).
int glds_tapmajor(const VmvGemmParams& p) {
    static int v = -2;
    if (v == -2) { const char* e = getenv("VMV_GLDS_TAPMAJOR"); v = e ? atoi(e) : -1; }
    return v >= 0 ? v : (p.M >= 16384 ? 1 : 0);
}

// rows of the tile group that shares W slices in an XCD's L2 (gemm_xglds.hip xglds_group_m; `conc` = blocks an XCD runs at once:
// 32 CUs x 1 or 2 blocks).  VMV_GLDS_GM forces it (A/B; 1 = the round-5 order).
int glds_group_m(int tiles_m, int tiles_n, int BM, int BN, int conc) {
    static int env = -2;
    if (env == -2) { const char* e = getenv("VMV_GLDS_GM"); env = e ? atoi(e) : -1; }
    if (env >= 1) return env;
    if (tiles_n < 2 || tiles_m < 2) return 1;
    int best = 1, best_cost = BM + conc * BN;
    for (int gm = 2; gm <= conc; gm *= 2) {
        const int gn = (conc + gm - 1) / gm;
        if (gn > tiles_n || gm > tiles_m) continue;
        const int cost = gm * BM + gn * BN;
        if (cost < best_cost) { best = gm; best_cost = cost; }
    }
    return best;
}

template <int WMW, int WN, int STAGES, bool PP = false>
int launch_glds(const VmvGemmParams& p, int total_steps, hipStream_t st) {
    using Cfg = GlCfg<WMW, WN, STAGES>;
    static_assert(Cfg::LPT >= 6 && Cfg::LPT <= 9 && (STAGES == 2 || 2 * Cfg::LPT <= 14), "wait_vmcnt literals");
    static_assert(Cfg::BM * (Cfg::BN * 2 + 16) <= Cfg::LDS_BYTES, "epilogue staging fits in the ring");
    const int tiles_m = (p.M + Cfg::BM - 1) / Cfg::BM;
    const int tiles_n = (p.N + Cfg::BN - 1) / Cfg::BN;
    const int ks = p.ksplit > 1 ? p.ksplit : 1;
    const int sps = (total_steps + ks - 1) / ks;
    static int ablate = -1;
    if (ablate < 0) { const char* e = getenv("VMV_GEMM_ABLATE"); ablate = e ? atoi(e) : 0; }
    dim3 grid(tiles_m * tiles_n, ks, 1);
    const int gm = glds_group_m(tiles_m, tiles_n, Cfg::BM, Cfg::BN, WMW == 2 ? 64 : 32);
    auto go = [&](auto tag) -> int {
        constexpr int AB = decltype(tag)::value;
        static std::atomic<unsigned long long> attr_set{0};
        if (const int rc_attr = vmv_lds_attr_once(attr_set, reinterpret_cast<const void*>(&gemm_glds_kernel<WMW, WN, STAGES, AB, PP>), Cfg::LDS_BYTES)) return rc_attr;
        VMV_LAUNCH((gemm_glds_kernel<WMW, WN, STAGES, AB, PP>), grid, dim3(Cfg::NT), Cfg::LDS_BYTES, st, p, tiles_m, tiles_n,
                           total_steps, sps, gm, glds_tapmajor(p));
        return VMV_OK;
    };
    int rc;
#if defined(VMV_EXPERIMENTS)       // (the ablation instantiations are not in the production library)
    switch (ablate) {
        case 1: rc = go(std::integral_constant<int, 1>{}); break;
        case 2: rc = go(std::integral_constant<int, 2>{}); break;
        case 3: rc = go(std::integral_constant<int, 3>{}); break;
        case 4: rc = go(std::integral_constant<int, 4>{}); break;
        case 5: rc = go(std::integral_constant<int, 5>{}); break;
        case 6: rc = go(std::integral_constant<int, 6>{}); break;
        default: rc = go(std::integral_constant<int, 0>{}); break;
    }
#else
    rc = go(std::integral_constant<int, 0>{});
#endif
    if (rc != VMV_OK) return rc;
    return vmv_launch_status();
}

}  // namespace

// Called by vmv_gemm (gemm.hip) after argument validation.  The split-K reduce pass stays in gemm.hip.
int vmv_gemm_glds_launch(const VmvGemmParams& p, int total_steps, int tile, hipStream_t st) {
    if (p.rowstat) return VMV_GLDS_UNSUPPORTED;          // LayerNorm-folded GEMMs: gemm_pglds.hip / gemm.hip epilogues only
    // 32-bit byte offsets through buffer descriptors: every operand must span < 2 GiB
    long maxrows = p.M;
    if (p.OH > 0) { const long src_rows = (long)(p.M / (p.OH * p.OW) + 1) * p.IH * p.IW; if (src_rows > maxrows) maxrows = src_rows; }
    for (int i = 0; i < p.nseg; ++i)
        if (maxrows * (long)p.seg[i].ld * 2 >= (1L << 31) - 65536) return VMV_GLDS_UNSUPPORTED;
    if ((long)p.N * p.ktot * 2 >= (1L << 31) - 65536) return VMV_GLDS_UNSUPPORTED;
    if (tile == VMV_TILE_PP256x128) return launch_glds<4, 4, 3, true>(p, total_steps, st);
    if (tile == VMV_TILE_PP256x160) {
        if (p.epilogue == VMV_EPI_GEGLU) return VMV_EINVAL;
        return launch_glds<4, 5, 3, true>(p, total_steps, st);
    }
    if (tile == VMV_TILE_256x128) return launch_glds<4, 4, 3>(p, total_steps, st);
    if (tile == VMV_TILE_G128x128) return launch_glds<2, 4, 2>(p, total_steps, st);
    if (tile == VMV_TILE_256x160 || tile == VMV_TILE_G128x160) {
        if (p.epilogue == VMV_EPI_GEGLU) return VMV_EINVAL;
        return tile == VMV_TILE_256x160 ? launch_glds<4, 5, 3>(p, total_steps, st) : launch_glds<2, 5, 2>(p, total_steps, st);
    }
    return VMV_EINVAL;
}

// gemm_sglds.hip — WAVE-SPECIALISED persistent variant of the LDS-DMA bf16 MFMA implicit GEMM (same contract as gemm.hip).
//
// Why: on gfx950 a wave that issues LDS-DMA (`buffer_load ... lds`) sits in its issue slot while the CU's texture path
// takes the wave-instructions one at a time (~21 cycles per 1-KB instruction, 118 GB/s per CU), and in a loop where the
// same waves also run the MFMAs that time does not overlap the matrix pipe — not as a burst after the chunk's barrier,
// and not spread one instruction per five MFMAs either.  tools/experiments/cu_overlap.hip, per 256x128x64 chunk and CU:
//     MFMA only 564 ns | DMA only 414 ns | same waves, burst 916 ns | same waves, interleaved 800 ns |
//     8 MFMA waves + 4 loader waves 695 ns (620 without the per-chunk barrier).
// So this kernel runs 12 waves per CU: waves 0-7 own the 256/192 x 128/160 accumulator tile exactly as in
// gemm_pglds.hip (fragment reads interleaved with the MFMAs, per-wave epilogue through a private LDS slab) and never
// touch global memory in the main loop; waves 8-11 do nothing but walk the K segments of the block's tile list and
// feed the 3-stage ring, up to three chunks ahead, across tile boundaries.  One s_barrier per chunk is the whole
// protocol:   loader: wait(chunk c+1 landed) -> B_c -> issue(chunk c+3 -> slot c % 3)
//             MFMA  : MFMAs(kk0) + reads(slot c, kk1) -> B_c -> MFMAs(kk1) + reads(slot c+1, kk0)
// Because the loader has registers to spare it handles every segment kind (im2col taps, temporal shifts, plain rows),
// so the conv GEMMs run on this kernel too.  All 12 waves get the same register allocation: <= 168 (3 waves per SIMD).
#include "gemm_glds_common.h"
#include <cstdlib>
#include <cstddef>
#include <type_traits>

using namespace vmvg;

namespace {

template <int WM_, int WN_>
struct SgCfg {
    static constexpr int NWC = 8, NL = 4, NT = 64 * (NWC + NL);
    static constexpr int S = 3;
    static constexpr int BM = 64 * WM_;                    // 4 wave rows x 16*WM
    static constexpr int BN = 32 * WN_;
    static constexpr int A_BYTES = BM * 128;
    static constexpr int W_BYTES = BN * 128;
    static constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
    static constexpr int RING_BYTES = S * STAGE_BYTES;
    static constexpr int PA = BM / 8 / NL;                 // A row groups (8 rows x 128 B = one wave-instruction) per loader wave
    static constexpr int PW = BN / 8 / NL;
    static constexpr int PER = PA + PW;                    // wave-instructions per loader wave per chunk
    static constexpr int STRIP = 16 * WN_ * 4;             // per MFMA wave: its bias values
    // epilogue slab per MFMA wave (16 rows x (columns * 2 + 16) B).  Where 160 KB allows it the slabs live behind the ring
    // and the loader never waits for an epilogue; otherwise they live in the ring slot the tile has just freed and the
    // loader holds that slot's refill until the epilogue barrier (see `dedicated` in the kernel).
    static constexpr int DED_SLAB = (WM_ == 3 && WN_ == 5) ? 2816 : (WN_ == 4 ? 1280 : 0);
    static constexpr int LDS_TOTAL = RING_BYTES + NWC * STRIP + NWC * DED_SLAB;
    static_assert((BM / 8) % NL == 0 && (BN / 8) % NL == 0, "row groups split evenly over the loader waves");
    static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");
    static_assert(NWC * 4096 <= STAGE_BYTES, "in-ring slabs fit in one ring slot");
};

template <int N> VMV_DEV void wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
    else if constexpr (N == 22) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
    else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if constexpr (N == 26) asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
    else static_assert(N == 0, "add the literal");
}

template <int WM, int WN, bool DBG>
__global__ __launch_bounds__(768, 3) void gemm_sglds_kernel(const VmvGemmParams p, const int tiles_n_, const int total_steps_, const int nitems_, const int fast_) {
    VMV_KERNEL_ENTER();
    // (kernel arguments are uniform, but once the segment table is indexed inside the loader's nested loops the compiler
    //  stops believing it and spills "divergent" loop bounds: pin them to SGPRs)
    const int tiles_n = __builtin_amdgcn_readfirstlane(tiles_n_);
    const int total_steps = __builtin_amdgcn_readfirstlane(total_steps_);
    const int nitems = __builtin_amdgcn_readfirstlane(nitems_);
    const bool fast = __builtin_amdgcn_readfirstlane(fast_) != 0;       // every segment is "separable" (see enter_segment)
    using Cfg = SgCfg<WM, WN>;
    constexpr int BN = Cfg::BN, BM = Cfg::BM, S = Cfg::S;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x;
    const int bid = blockIdx.x;
    const int my_items = bid < nitems ? (nitems - 1 - bid) / G + 1 : 0;
    if (my_items == 0) return;
    const int total = my_items * total_steps;              // chunks this block consumes, flattened over its tiles

    auto item_tile = [&](int v, int& m0, int& n0) {        // XCD-aware bijection item -> tile (see gemm_glds.hip)
        const int q = nitems >> 3, r = nitems & 7;
        const int xcd = v & 7, idx = v >> 3;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        const int tn = logical % tiles_n;
        m0 = (logical / tiles_n) * BM;
        n0 = tn * BN;
    };
    // DBG (VMV_GEMM_ABLATE=4): block 0 stamps s_memtime per chunk into p.workspace, uint64 [chunk < 64][8]:
    //   loader wave 8: 0 = before wait, 1 = after wait, 2 = after B_c, 3 = after issue;  MFMA wave 0: 4 = before B_c, 5 = after B_c
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(p.workspace);
    auto stamp = [&](int c, int k) {
        if constexpr (DBG) {
            if (bid == 0 && lane == 0 && c < 64 && stamps) stamps[c * 8 + k] = __builtin_readcyclecounter();
        }
    };
    const bool geglu = p.epilogue == VMV_EPI_GEGLU;
    const bool dedicated = Cfg::DED_SLAB >= 2816 || (Cfg::DED_SLAB > 0 && geglu);      // uniform over the grid

    if (wave >= Cfg::NWC) {
        // =============================================================== loader waves
        // The two MFMA waves it shares a SIMD with keep the vector issue port busy; a loader wave is asleep most of the time
        // (barrier / vmcnt), so give its few address VALU ops precedence — otherwise a tile switch (row gather set-up, ~200
        // VALU instructions) takes > 2.5k cycles and drains the ring.
        __builtin_amdgcn_s_setprio(3);
        const int lw = wave - Cfg::NWC;
        const int lrow = lane >> 3;
        // row r of a tile keeps logical 16-B slot s at physical slot s ^ ((r >> 1) & 7); r = 8 g + lrow with g = lw + NL k
        const int lsw = (lane & 7) ^ ((((lw & 1) << 2) + (lane >> 4)) & 7);
        const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, SRD_RECORDS, SRD_FLAGS);
        RowInfo rinfo[Cfg::PA];
        uint32_t avo[Cfg::PA], wvo[Cfg::PW];
        uint32_t edge[Cfg::PA];                  // fast path: bit set = the row sits ON that border (1 top, 2 bottom, 4 left, 8 right,
                                                 // 16 first frame, 32 last frame) and the tap that crosses it reads zeros
        int seg_dlo = 0, seg_dhi = 0;            // fast path: byte offset of the segment's tap (64-bit, kept as two pinned
                                                 // SGPRs: as a plain `long` it is treated as divergent and every LDS-DMA
                                                 // gets a waterfall loop), folded into the descriptor base
        int s = 0, kc = 0, koff = 0, islot = 0, L_item = bid, L_left = 0;
        // The segment table is read straight from the kernel-argument segment (scalar loads): indexing the by-value struct
        // with a run-time index makes the compiler copy all of `p` to scratch.
        typedef const __attribute__((address_space(4))) VmvGemmSeg* SegPtr;
        const SegPtr segs = (SegPtr)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() +
                                     offsetof(VmvGemmParams, seg));
        auto load_seg = [&](int i) __attribute__((always_inline)) -> VmvGemmSeg {
            VmvGemmSeg g;
            g.src = segs[i].src; g.ld = segs[i].ld; g.k = segs[i].k; g.mode = segs[i].mode; g.d0 = segs[i].d0; g.d1 = segs[i].d1;
            g._pad = 0;
            return g;
        };
        // Segment switch.  In the UNet every K segment is "separable": a plain row, a 3x3 tap of a stride-1 same-size conv, or
        // a +-1 frame shift — the source row of output row m is m itself plus a per-SEGMENT constant, and whether the tap
        // falls off the image depends only on which borders the row touches.  Then a switch costs four VALU ops per row
        // group (the loader shares its SIMD's issue port with two MFMA waves: the general gather below, ~30 VALU ops per
        // row group with its bounds checks, stalled the ring for ~2.3k cycles at every tap = every 5 chunks at C = 320).
        auto enter_segment = [&]() __attribute__((always_inline)) {
            const VmvGemmSeg sg = load_seg(__builtin_amdgcn_readfirstlane(s));
            if (fast) {
                uint32_t need = 0;
                long delta = 0;
                if (sg.mode == VMV_SEG_SPATIAL) {
                    need = (sg.d0 < 0 ? 1u : 0u) | (sg.d0 > 0 ? 2u : 0u) | (sg.d1 < 0 ? 4u : 0u) | (sg.d1 > 0 ? 8u : 0u);
                    delta = (long)(sg.d0 * p.IW + sg.d1) * sg.ld;
                } else if (sg.mode == VMV_SEG_TEMPORAL) {
                    need = (sg.d0 < 0 ? 16u : 0u) | (sg.d0 > 0 ? 32u : 0u);
                    delta = (long)sg.d0 * p.P * sg.ld;
                }
                seg_dlo = __builtin_amdgcn_readfirstlane((int)((delta * 2) & 0xffffffffL));
                seg_dhi = __builtin_amdgcn_readfirstlane((int)((delta * 2) >> 32));
#pragma unroll
                for (int k = 0; k < Cfg::PA; ++k)
                    avo[k] = (rinfo[k].m >= 0 && (edge[k] & need) == 0u) ? (uint32_t)(rinfo[k].m * sg.ld + lsw * 8) * 2u : OOB;
                return;
            }
#pragma unroll
            for (int k = 0; k < Cfg::PA; ++k) {
                const int off = seg_row_offset(p, sg, rinfo[k]);
                avo[k] = off >= 0 ? (uint32_t)(off + lsw * 8) * 2u : OOB;
            }
        };
        auto setup_item = [&](int v) __attribute__((always_inline)) {
            int m0, n0;
            item_tile(v, m0, n0);
#pragma unroll
            for (int k = 0; k < Cfg::PA; ++k) {
                const int m = m0 + (lw + Cfg::NL * k) * 8 + lrow;
                RowInfo r;
                r.m = (m < p.M) ? m : -1;
                r.nb = 0; r.oy = 0; r.ox = 0; r.fr = 0;
                if (p.OH > 0) {
                    const int hw = p.OH * p.OW;
                    const int n = m / hw, rem = m - n * hw;
                    r.nb = n * p.IH * p.IW;
                    r.oy = rem / p.OW;
                    r.ox = rem - r.oy * p.OW;
                }
                if (p.P > 0) r.fr = (m / p.P) % p.F;
                rinfo[k] = r;
                edge[k] = (p.OH > 0 ? (r.oy == 0 ? 1u : 0u) | (r.oy == p.OH - 1 ? 2u : 0u) | (r.ox == 0 ? 4u : 0u) | (r.ox == p.OW - 1 ? 8u : 0u) : 0u) |
                          (p.P > 0 ? (r.fr == 0 ? 16u : 0u) | (r.fr == p.F - 1 ? 32u : 0u) : 0u);
            }
#pragma unroll
            for (int j = 0; j < Cfg::PW; ++j) {
                const int n = n0 + (lw + Cfg::NL * j) * 8 + lrow;
                wvo[j] = (n < p.N) ? (uint32_t)(n * p.ktot + lsw * 8) * 2u : OOB;
            }
            s = 0; kc = 0; koff = 0;
            L_left = total_steps;
            enter_segment();
        };
        auto issue_chunk = [&]() __attribute__((always_inline)) {               // LDS-DMA this wave's PER row groups of the next chunk into ring slot `islot`
            const VmvGemmSeg sg = load_seg(__builtin_amdgcn_readfirstlane(s));
            const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(reinterpret_cast<const char*>(sg.src)) +
                    (((long)__builtin_amdgcn_readfirstlane(seg_dhi) << 32) | (long)(uint32_t)__builtin_amdgcn_readfirstlane(seg_dlo)),
                0, SRD_RECORDS, SRD_FLAGS);
            const bool kall = (kc + BK) <= sg.k || (kc + lsw * 8) < sg.k;       // K tail of a segment: zero fill
            unsigned char* abase = smem + islot * Cfg::STAGE_BYTES + lw * 1024;
            unsigned char* wbase = abase + Cfg::A_BYTES;
            const uint32_t a_so = (uint32_t)kc * 2u, w_so = (uint32_t)(koff + kc) * 2u;
#pragma unroll
            for (int k = 0; k < Cfg::PA; ++k) VMV_BLDS16(a_rsrc, abase + k * (Cfg::NL * 1024), kall ? avo[k] : OOB, a_so);
#pragma unroll
            for (int j = 0; j < Cfg::PW; ++j) VMV_BLDS16(w_rsrc, wbase + j * (Cfg::NL * 1024), kall ? wvo[j] : OOB, w_so);
            islot = islot + 1 == S ? 0 : islot + 1;
            kc += BK;
            --L_left;
            if (kc >= sg.k) {
                koff += sg.k; ++s; kc = 0;
                if (L_left > 0) enter_segment();
            }
            if (L_left == 0) {
                L_item += G;
                if (L_item < nitems) setup_item(L_item);
            }
        };
        setup_item(L_item);
        int issued = 0;
        const int pro = total < S ? total : S;
        for (int i = 0; i < pro; ++i) { issue_chunk(); ++issued; }
        if (pro == 3) wait_vm<2 * Cfg::PER>(); else if (pro == 2) wait_vm<Cfg::PER>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();                       // X0: chunk 0 is visible
        int t = 0;                                          // chunk index inside the tile being consumed
        for (int c = 0; c < total; ++c) {
            if (lw == 0) stamp(c, 0);
            if (c + 1 < total) { if (issued >= c + 3) wait_vm<Cfg::PER>(); else wait_vm<0>(); }
            if (lw == 0) stamp(c, 1);
            __builtin_amdgcn_s_barrier();                   // B_c: slot c % 3 has been read by everyone, chunk c + 1 is visible
            if (lw == 0) stamp(c, 2);
            if (++t == total_steps) {
                t = 0;
                if (!dedicated) __builtin_amdgcn_s_barrier();          // E: the epilogue has left its slabs in slot c % 3
            }
            if (issued < total) { issue_chunk(); ++issued; }
            if (lw == 0) stamp(c, 3);
        }
        wait_vm<0>();
        return;
    }

    // =================================================================== MFMA waves
    const int wave_m = wave >> 1, wave_n = wave & 1;
    f32x4_t acc[WN][WM];
    auto zero_acc = [&]() {
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int i = 0; i < WM; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
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
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int i = 0; i < WM; ++i)
                acc[j][i] = VMV_MFMA16(wf[j], af[i], acc[j][i], 0, 0, 0);
    };
    // one MFMA phase: the WM*WN MFMAs on (af, wf) with the WM+WN fragment reads of the next phase spread between them
    constexpr int NM = WM * WN, NRD = WM + WN;
    auto phase = [&](const elem8_t (&af)[WM], const elem8_t (&wf)[WN], elem8_t (&afn)[WM], elem8_t (&wfn)[WN],
                     const int slot_n, const int kk_n) {
        const u32x4_t* a = reinterpret_cast<const u32x4_t*>(smem + slot_n * Cfg::STAGE_BYTES) + (wave_m * 16 * WM + frow) * 8;
        const u32x4_t* w = reinterpret_cast<const u32x4_t*>(smem + slot_n * Cfg::STAGE_BYTES + Cfg::A_BYTES) +
                           (wave_n * 16 * WN + frow) * 8;
        const int slot = (kk_n * 4 + fgrp) ^ fswz;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int j = m / WM, i = m % WM;
            acc[j][i] = VMV_MFMA16(wf[j], af[i], acc[j][i], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < NRD; ++r)
                if (((2 * r + 1) * NM) / (2 * NRD) == m) {
                    if (r < WM) afn[r < WM ? r : 0] = __builtin_bit_cast(elem8_t, a[(r < WM ? r : 0) * 16 * 8 + slot]);
                    else wfn[r >= WM ? r - WM : 0] = __builtin_bit_cast(elem8_t, w[(r >= WM ? r - WM : 0) * 16 * 8 + slot]);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ------------------------------------------------------------------ per-wave epilogue through a private LDS slab
    // (as gemm_pglds.hip: bias / residual loads issued before the tile's last MFMAs, 16-row groups transposed through
    //  the slab, whole 16-byte lanes stored through buffer descriptors; see the comments there)
    const int N_out = geglu ? p.N / 2 : p.N;
    const bool staged = !p.out_fp32 && (p.ldo & 7) == 0 && (N_out & 7) == 0 && vmv_ptr_aligned16(p.out) &&
                        (!p.residual || ((p.ldr & 7) == 0 && vmv_ptr_aligned16(p.residual)));
    constexpr int OWC_MAX = 16 * WN;
    constexpr int NR_MAX = (16 * (OWC_MAX / 8) + 63) / 64;
    float* bias_lds = reinterpret_cast<float*>(smem + Cfg::RING_BYTES + wave * Cfg::STRIP);
    u32x4_t resv[NR_MAX];                                  // residual of the NEXT 16-row group (one group ahead)
    u32x4_t sd_prev[NR_MAX];                               // store data of the last 16-row group written (see the epilogue)
#pragma unroll
    for (int r = 0; r < NR_MAX; ++r) sd_prev[r] = u32x4_t{0u, 0u, 0u, 0u};
    f32x4_t bias_hold = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, SRD_RECORDS, SRD_FLAGS);
    const __amdgpu_buffer_rsrc_t res_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.residual), 0, SRD_RECORDS, SRD_FLAGS);
    const bool has_res = staged && p.residual != nullptr;
    const float res_scale = p.res_scale != 0.f ? p.res_scale : 1.f;
    // (`le` = the lane id passed through an opaque asm at the top of every tile's epilogue: with only 168 registers the
    //  compiler otherwise hoists ~35 loop-invariant per-lane offsets / masks out of the tile loop, spills them, and reloads
    //  them from scratch next to every store — and a scratch reload's vmcnt wait drains the stores and residual loads in
    //  flight: the epilogue measured 26k cycles per 192x160 tile that way, against ~7k in gemm_pglds.hip)
    auto unit_offsets = [&](int le, int m0, int n0, int i, int r, int ld, auto geglu_tag) -> uint32_t {
        constexpr bool GEGLU = decltype(geglu_tag)::value;
        constexpr int OWC = GEGLU ? 8 * WN : 16 * WN;
        constexpr int UW = OWC / 8;
        constexpr int NU = 16 * UW;
        const int unit = le + 64 * r;
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
        int le = lane;
        asm volatile("" : "+v"(le));
        {
            const int n = n0 + wave_n * 16 * WN + 4 * le;
            bias_hold = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (p.bias && le < 4 * WN && n < p.N) bias_hold = *reinterpret_cast<const f32x4_t*>(p.bias + n);
        }
        if (has_res) {
            const uint32_t sb = group_base(m0, n0, 0, p.ldr, geglu_tag);
#pragma unroll
            for (int r = 0; r < NR; ++r)
                resv[r] = __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, unit_offsets(le, m0, n0, 0, r, p.ldr, geglu_tag), sb, 0);
        }
    };
    auto epilogue = [&](int m0, int n0, unsigned char* slab, auto geglu_tag) {
        constexpr bool GEGLU = decltype(geglu_tag)::value;
        int le = lane;
        asm volatile("" : "+v"(le));
        const int frow = le & 15, fgrp = le >> 4;             // (shadow the kernel-scope copies: see unit_offsets)
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
        if (le < 4 * WN) *reinterpret_cast<f32x4_t*>(bias_lds + 4 * le) = bias_hold;
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int m = mbase + 16 * i;
            f32x4_t rv[WN];
            if (p.rowvec) {
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
            u32x2_t packed[WN];
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                if (GEGLU && (j & 1)) continue;
                f32x4_t v = acc[j][i] + *reinterpret_cast<const f32x4_t*>(bias_lds + 16 * j + 4 * fgrp);
                if constexpr (GEGLU) {
                    if constexpr ((WN & 1) == 0) {
                        const f32x4_t g = acc[(j + 1) % WN][i] +
                                          *reinterpret_cast<const f32x4_t*>(bias_lds + 16 * ((j + 1) % WN) + 4 * fgrp);
                        v.x *= gelu_erf_f(g.x); v.y *= gelu_erf_f(g.y); v.z *= gelu_erf_f(g.z); v.w *= gelu_erf_f(g.w);
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
                *reinterpret_cast<u32x2_t*>(slab + frow * RB + tc * 2) = packed[j];
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            asm volatile("" ::: "memory");
            const uint32_t sb = group_base(m0, n0, i, p.ldo, geglu_tag);
            // all slab reads of the group first, THEN its stores (gfx950: a buffer_store's data registers must not be the
            // target of the next ds_read — see gemm_pglds.hip)
            u32x4_t vout[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int unit = le + 64 * r;
                const int rr = unit / UW, u = unit - rr * UW;
                vout[r] = u32x4_t{0u, 0u, 0u, 0u};
                if (unit < NU) vout[r] = *reinterpret_cast<const u32x4_t*>(slab + rr * RB + u * 16);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_sched_barrier(0);
            // Store-data discipline.  A buffer_store may read its data registers late when the store path is backed up, and
            // an LDS read that RETURNS into those registers meanwhile corrupts the store (seen in gemm_pglds.hip; with 168
            // registers the allocator reuses them at once).  So (1) store data is always a VALU-written copy (`sd`), never an
            // LDS-read destination, and (2) the previous group's `sd` is kept alive (a fake use) until this group's LDS reads
            // — bias strip, slab — have all returned: no LDS read issued within a whole group after a store can be given
            // the store's registers.  VMEM loads are ordered behind the store in the same queue and need no such care.
#pragma unroll
            for (int r = 0; r < NR; ++r)
                if (i > 0) asm volatile("" ::"v"(sd_prev[r]));          // (group 0: the previous tile's last group was retired
                                                                         //  after this tile's first fragment reads)
            u32x4_t sd[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                u32x4_t v = vout[r];
                if (has_res) {
                    float a[8], b[8];
                    unpack8(v, a); unpack8(resv[r], b);
#pragma unroll
                    for (int e = 0; e < 8; ++e) a[e] += res_scale * b[e];
                    v = pack8(a);
                }
                asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                             : "=&v"(sd[r].x), "=&v"(sd[r].y), "=&v"(sd[r].z), "=&v"(sd[r].w)
                             : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
            }
            // The next group's residual is requested BEFORE this group's stores: vmcnt retires in order, so a load issued
            // behind stores can only be waited for together with them (a full write round trip per group).
            asm volatile("" ::: "memory");
            if (has_res && i + 1 < WM) {
                const uint32_t sr = group_base(m0, n0, i + 1, p.ldr, geglu_tag);
#pragma unroll
                for (int r = 0; r < NR; ++r)
                    resv[r] = __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, unit_offsets(le, m0, n0, i + 1, r, p.ldr, geglu_tag), sr, 0);
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int r = 0; r < NR; ++r)
                __builtin_amdgcn_raw_buffer_store_b128(sd[r], out_rsrc, unit_offsets(le, m0, n0, i, r, p.ldo, geglu_tag), sb, 0);
#pragma unroll
            for (int r = 0; r < NR; ++r) sd_prev[r] = sd[r];
            asm volatile("" ::: "memory");
        }
    };

    // ------------------------------------------------------------------ chunk pipeline of the MFMA waves
    elem8_t a0[WM], w0[WN], a1[WM], w1[WN];
    int st = 0;                                            // ring slot of the chunk being consumed
    int cdbg = 0;
    __builtin_amdgcn_s_barrier();                          // X0
    asm volatile("" ::: "memory");
    for (int item = bid; item < nitems; item += G) {
        int m0, n0;
        item_tile(item, m0, n0);
        read_frags(st, 0, a0, w0);
        // (the previous tile's last store data stays alive until these fragment reads have returned: see the epilogue)
        __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
        for (int r = 0; r < NR_MAX; ++r) asm volatile("" ::"v"(sd_prev[r]));
        for (int t = 0; t + 1 < total_steps; ++t) {
            phase(a0, w0, a1, w1, st, 1);
            __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): my reads of slot st are done
            if (wave == 0) stamp(cdbg, 4);
            __builtin_amdgcn_s_barrier();                  // B_c
            if (wave == 0) stamp(cdbg, 5);
            if constexpr (DBG) ++cdbg;
            asm volatile("" ::: "memory");
            const int stn = st + 1 == S ? 0 : st + 1;
            phase(a1, w1, a0, w0, stn, 0);
            st = stn;
        }
        // last chunk of the tile
        phase(a0, w0, a1, w1, st, 1);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        if (wave == 0) stamp(cdbg, 4);
        __builtin_amdgcn_s_barrier();                      // B_c (last of the tile): slot st may be reused
        if (wave == 0) stamp(cdbg, 5);
        if constexpr (DBG) ++cdbg;
        asm volatile("" ::: "memory");
        // bias + residual start their round trip under the tile's last MFMAs (a0 / w0 are dead here: issuing the prefetch
        // one phase earlier costs 20-28 more live registers exactly where the kernel peaks, and spills)
        if (geglu) epilogue_prefetch(m0, n0, std::true_type{}); else epilogue_prefetch(m0, n0, std::false_type{});
        asm volatile("" ::: "memory");
        mma(a1, w1);
        unsigned char* slab = dedicated ? smem + Cfg::RING_BYTES + Cfg::NWC * Cfg::STRIP + wave * Cfg::DED_SLAB
                                        : smem + st * Cfg::STAGE_BYTES + wave * 4096;
        st = st + 1 == S ? 0 : st + 1;
        if (geglu) epilogue(m0, n0, slab, std::true_type{}); else epilogue(m0, n0, slab, std::false_type{});
        zero_acc();
        if (!dedicated) {
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();                  // E: slabs released, the loader may refill the slot
            asm volatile("" ::: "memory");
        }
    }
}

template <int WM, int WN>
int launch_sglds(const VmvGemmParams& p, int total_steps, hipStream_t st) {
    using Cfg = SgCfg<WM, WN>;
    const int tiles_m = (p.M + Cfg::BM - 1) / Cfg::BM;
    const int tiles_n = (p.N + Cfg::BN - 1) / Cfg::BN;
    const int nitems = tiles_m * tiles_n;
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0, n = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return (int)e;
        if (n < 8) n = 8;
        ncu = n & ~7;                                // whole XCD groups: item & 7 == block & 7 in every round
    }
    const int G = nitems < ncu ? nitems : ncu;
    int fast = 1;                                    // all segments separable (see the loader's enter_segment)?
    for (int i = 0; i < p.nseg; ++i) {
        const VmvGemmSeg& sg = p.seg[i];
        if (sg.mode == VMV_SEG_SPATIAL)
            fast = fast && p.stride == 1 && p.ups == 0 && p.OH == p.IH && p.OW == p.IW && sg.d0 >= -1 && sg.d0 <= 1 && sg.d1 >= -1 && sg.d1 <= 1;
        else if (sg.mode == VMV_SEG_TEMPORAL)
            fast = fast && sg.d0 >= -1 && sg.d0 <= 1;
    }
    { static int nofast = -1; if (nofast < 0) { const char* e = getenv("VMV_GEMM_NOFAST"); nofast = e ? atoi(e) : 0; } if (nofast) fast = 0; }
    static int ablate = -1;
    if (ablate < 0) { const char* e = getenv("VMV_GEMM_ABLATE"); ablate = e ? atoi(e) : 0; }
    auto go = [&](auto dbg_tag) -> int {
        constexpr bool DBG = decltype(dbg_tag)::value;
        static std::atomic<unsigned long long> attr_set{0};
        if (const int rc_attr = vmv_lds_attr_once(attr_set, reinterpret_cast<const void*>(&gemm_sglds_kernel<WM, WN, DBG>), Cfg::LDS_TOTAL)) return rc_attr;
        hipLaunchKernelGGL((gemm_sglds_kernel<WM, WN, DBG>), dim3(G), dim3(Cfg::NT), Cfg::LDS_TOTAL, st, p, tiles_n, total_steps, nitems,
                           fast);
        return VMV_OK;
    };
    const int rc = ablate == 4 ? go(std::true_type{}) : go(std::false_type{});
    if (rc != VMV_OK) return rc;
    return vmv_launch_status();
}

}  // namespace

// Called by vmv_gemm (gemm.hip) after argument validation; split-K shapes stay on the non-persistent kernels.
int vmv_gemm_sglds_launch(const VmvGemmParams& p, int total_steps, int tile, hipStream_t st) {
    if (p.ksplit > 1 || p.rowstat) return VMV_GLDS_UNSUPPORTED;
    long maxrows = p.M;
    if (p.OH > 0) { const long src_rows = (long)(p.M / (p.OH * p.OW) + 1) * p.IH * p.IW; if (src_rows > maxrows) maxrows = src_rows; }
    for (int i = 0; i < p.nseg; ++i)
        if (maxrows * (long)p.seg[i].ld * 2 >= (1L << 31) - 65536) return VMV_GLDS_UNSUPPORTED;
    if ((long)p.N * p.ktot * 2 >= (1L << 31) - 65536) return VMV_GLDS_UNSUPPORTED;
    // the epilogue addresses out / residual through buffer descriptors too (32-bit byte offsets)
    if ((long)(p.M + 256) * p.ldo * (p.out_fp32 ? 4 : 2) >= (1L << 31) - 65536) return VMV_GLDS_UNSUPPORTED;
    if (p.residual && (long)(p.M + 256) * p.ldr * 2 >= (1L << 31) - 65536) return VMV_GLDS_UNSUPPORTED;
    if (tile == VMV_TILE_S256x128) return launch_sglds<4, 4>(p, total_steps, st);
    if (p.epilogue == VMV_EPI_GEGLU) return VMV_EINVAL;
    if (tile == VMV_TILE_S192x160) return launch_sglds<3, 5>(p, total_steps, st);
    if (tile == VMV_TILE_S256x160) return launch_sglds<4, 5>(p, total_steps, st);
    return VMV_EINVAL;
}

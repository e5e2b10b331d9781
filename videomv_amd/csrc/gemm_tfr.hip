// gemm_tfr.hip — FRAME-RESIDENT temporal (3,1,1) convolution with the GroupNorm apply (+ SiLU) on its A path (same contract as
// gemm.hip / vmv.h; TemporalConvBlock_v2, tools/modules/unet/util.py:1357-1392: 4 x [GroupNorm over all frames -> SiLU -> Conv3d(3,1,1)]).
//
// Why (lead item).  In the tile kernels the three taps of a temporal convolution are three K segments: a tile of 256
// consecutive rows streams the same activation tensor through LDS THREE times (rows m - P, m, m + P), and in front of every one of the
// 88 temporal convolutions of a forward sits a GroupNorm apply pass that reads and rewrites the whole tensor (31 us at the first level
// for a 95-us convolution).  Here a block owns ALL F frames of PT pixels of one sample (F x PT <= 192 rows = twelve 16-row MFMA
// fragments, LDS row = f * PT + j) and all-or-a-320-wide-slice of the output channels:
//   * A is staged ONCE per 64-channel chunk and there, in LDS, the norm's per-(sample, channel) scale / shift from
//     vmv_groupnorm_table and the SiLU are applied, once per element — x is read raw, the normalised tensor is never written or
//     re-read, only the norm's statistics pass is left;
//   * the three taps are ROW-SHIFTED VIEWS of that one LDS tile: tap dt multiplies the fragment whose lane-row r reads LDS row
//     r + dt * PT; frames -1 and F are PT zero rows in front of / behind the tile (the frame axis is whole: no halo exchange, no
//     border select) — A bytes through the CU per MAC / 3, and the B-fragment swizzle is the same for all six row tiles of a wave
//     because 16-row steps change (row >> 1) by 8;
//   * W streams through a four-stage ring of (tap, 32-channel) stages of 320 rows x 64 B by LDS-DMA, exactly the W side of
//     gemm_xglds.hip (slot swizzle T[(row >> 2) & 3]); one block barrier per stage = per 30 MFMAs of a wave.
// 8 waves = 4 (rows) x 2 (columns); a wave owns 48 rows x 160 columns = 3 x 10 accumulator tiles (120 registers); a stage is two
// half-phases of 15 MFMAs (column half h), the fragments of the next half (5 W, + the 3 row-shifted B fragments when the stage
// changes) are read before the MFMAs of the current one — gemm_xglds.hip's schedule; a 6 x 5 wave tile with whole-stage double
// buffering (88 fragment registers) did not fit beside the accumulators (60-80 spilled registers, each reload a vmcnt(0)).  Per 32-channel k-step a CU moves 60 KB of W by DMA + 12 KB of A through registers for 11.8 MFLOP
// (161 FLOP per byte against 142 for the 256 x 320 tile, with A no longer on the DMA path).
// A-path schedule (in-order vmcnt): the three 16-byte units a lane owns of A stage a + 1 go out by LDS-DMA at the end of phase 6 a - 1
// into the other A buffer, are old enough to be covered by the ring's own counted wait at phase 6 a + 2, are normalised in place by the
// lane that fetched them during the MFMAs of phases 6 a + 2 .. 6 a + 4, and become visible with the barrier of phase 6 a + 5.
#include "gemm_glds_common.h"
#include <cstdlib>
#include <type_traits>

using namespace vmvg;

namespace {

constexpr int TF_ROWS = 192;                     // rows of a tile (F x PT <= 192)
constexpr int TF_PAD = 16;                       // zero rows in front of / behind the tile (PT <= 16)
constexpr int TF_AROWS = TF_ROWS + 2 * TF_PAD;   // 224
constexpr int TF_BN = 320;
constexpr int TF_NW = 8, TF_NT = 512;
constexpr int TF_S = 4;                          // W ring stages
constexpr int TF_WBYTES = TF_BN * 64;            // one W stage: 320 rows x 32 channels
constexpr int TF_ABYTES = TF_AROWS * 128;        // one A buffer: 224 rows x 64 channels
constexpr int TF_OFF_A = TF_S * TF_WBYTES;                       // 81920
constexpr int TF_OFF_TAB = TF_OFF_A + 2 * TF_ABYTES;             // 139264
constexpr int TF_MAXC = 1280;
constexpr int TF_OFF_BIAS = TF_OFF_TAB + 2 * TF_MAXC * 4;        // 149504
constexpr int TF_LDS = TF_OFF_BIAS + TF_BN * 4;                  // 150784
constexpr int TF_ROWB = TF_BN * 2 + 16;                          // epilogue staging row: 656 B
static_assert(TF_ROWS * TF_ROWB <= TF_OFF_TAB && TF_LDS <= 160 * 1024, "LDS budget");
constexpr int TF_WM = 3, TF_WN = 10, TF_WH = 5;  // accumulator tiles per wave: 3 row fragments x 10 column tiles, in two halves of 5
constexpr int TF_NWI = 2, TF_NWX = 4;            // W wave-instructions per wave per stage: 2, + 1 for waves < 4 (20 groups of 16 rows)

#ifndef VMV_TFR_SGB
#define VMV_TFR_SGB 0      // experiments: 1 = pin "1 MFMA : 2 VALU" with sched_group_barrier in the half-phases that carry an A half-unit
#endif

template <bool GN>
__global__ __launch_bounds__(512, 1) void gemm_tfr_kernel(const VmvGemmParams p, const int tiles_g, const int tiles_n, const int PT) {
    VMV_KERNEL_ENTER();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int C = p.seg[0].k, F = p.F, P = p.P;
    const int NA = C >> 6;                                      // 64-channel A stages = blocks of six phases (2 k-steps x 3 taps)

    // ---- XCD-aware tile mapping (bijective, as gemm_xglds.hip): the N tiles of one row tile are adjacent
    const int nblk = gridDim.x;
    int logical;
    {
        const int bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int rt = logical / tiles_n, n0 = (logical - rt * tiles_n) * TF_BN;
    const int b = rt / tiles_g, px0 = (rt - b * tiles_g) * PT;
    const int RV = F * PT;                                      // valid rows of the tile (<= 192)
    const long mb = (long)b * F * P + px0;                      // global row of (frame 0, pixel px0)

    unsigned char* const abuf = smem + TF_OFF_A;
    float* const tab = reinterpret_cast<float*>(smem + TF_OFF_TAB);          // [2][C]: scale, shift of this sample's norm
    float* const bias_lds = reinterpret_cast<float*>(smem + TF_OFF_BIAS);

    // ---- A loader: LDS-DMA straight into the NEXT A buffer (no staging registers), then — GroupNorm fold only — an in-place pass
    //      over the lane's own three 16-byte units (ds_read -> scale / shift -> SiLU -> ds_write): a lane transforms exactly the bytes
    //      its own DMA wrote, so its own vmcnt wait is all the ordering that pass needs.  A wave instruction covers 8 rows x 128 B;
    //      lane -> (row lane >> 3, physical slot lane & 7), logical k-slot (lane & 7) ^ ((row >> 1) & 7), applied to the SOURCE
    //      address (the LDS image of a DMA is lane-linear).  Unit q of this lane = tile row (tid >> 3) + 64 q; LDS row = that + 16,
    //      and 64-row steps leave (row >> 1) & 7 unchanged: one swizzle, one LDS offset (+ 8192 q).
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.seg[0].src), 0, SRD_RECORDS, SRD_FLAGS);
    const int aslot = (tid & 7) ^ ((((tid >> 3) + TF_PAD) >> 1) & 7);        // logical slot (8 channels) this lane's units hold
    uint32_t avo[3];        // byte offset of the unit's source (OOB: rows outside the tile / the image -> the DMA writes zeros)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int r = (tid >> 3) + 64 * q;
        const int f = r / PT, j = r - f * PT;
        const bool ok = r < RV && px0 + j < P;
        avo[q] = ok ? (uint32_t)(((mb + (long)f * P + j) * p.seg[0].ld + aslot * 8) * 2) : OOB;
    }
    const uint32_t ado = (uint32_t)(TF_PAD * 128 + tid * 16);               // this lane's unit 0 inside an A buffer
    auto a_request = [&](const int a) {          // the three units of A stage a (channels 64 a ..) into buffer a & 1
        unsigned char* dst = abuf + (a & 1) * TF_ABYTES + TF_PAD * 128 + wave * 1024;
#pragma unroll
        for (int q = 0; q < 3; ++q) VMV_BLDS16(a_rsrc, dst + q * 8192, avo[q], (uint32_t)a * 128u);
    };
    // half-unit hu = 2 q + h (4 channels = 8 bytes) of stage a, in place: elem(silu(x * scale + shift)); zero rows stay zero.  Six of
    // them per A stage, one per MFMA half-phase (below), so that each is ~30 VALU operations beside 15 MFMAs.
    auto a_commit_half = [&](const int a, const int hu) {
        if constexpr (GN) {
            const int q = hu >> 1, h = hu & 1;
            // (the two lane-dependent bases are made opaque per use: left visible, the six half-units of the unrolled block each get
            //  their own pre-added address registers hoisted out of the loop — 12 registers the kernel does not have)
            uint32_t ad = ado, ts8 = (uint32_t)aslot * 32u;
            asm volatile("" : "+v"(ad), "+v"(ts8));
            u32x2_t* up = reinterpret_cast<u32x2_t*>(abuf + (a & 1) * TF_ABYTES + ad + q * 8192 + h * 8);
            const u32x2_t v = *up;
            const float* t4 = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(tab) + a * 256 + ts8 + h * 16);
            const f32x4_t sc = *reinterpret_cast<const f32x4_t*>(t4), sh = *reinterpret_cast<const f32x4_t*>(t4 + C);
            float x0 = fmaf(elem_lo(v.x), sc.x, sh.x), x1 = fmaf(elem_hi(v.x), sc.y, sh.y);
            float x2 = fmaf(elem_lo(v.y), sc.z, sh.z), x3 = fmaf(elem_hi(v.y), sc.w, sh.w);
            if (p.gn_silu) { x0 = silu_f(x0); x1 = silu_f(x1); x2 = silu_f(x2); x3 = silu_f(x3); }
            u32x2_t o = u32x2_t{pack_elem2(x0, x1), pack_elem2(x2, x3)};
            if (avo[q] == OOB) o = u32x2_t{0u, 0u};               // rows outside the tile stay the zero padding of the frame axis
            *up = o;
        }
    };
    auto a_commit = [&](const int a, const int q) { a_commit_half(a, 2 * q); a_commit_half(a, 2 * q + 1); };

    // ---- W loader (gemm_xglds.hip): a wave instruction covers 16 rows x 64 B; lane -> (row lane >> 2, physical slot lane & 3),
    //      logical k-slot (lane & 3) ^ T[(row >> 2) & 3], T = {0, 2, 3, 1}
    const int lrow = lane >> 2;
    const int lsw = (lane & 3) ^ ((0x78 >> (2 * ((lane >> 4) & 3))) & 3);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, (uint32_t)p.N * (uint32_t)p.ktot * 2u, SRD_FLAGS);
    const uint32_t wvo0 = (uint32_t)((n0 + wave * 16 + lrow) * p.ktot + lsw * 8) * 2u;
    const uint32_t wstride = (uint32_t)(TF_NW * 16 * p.ktot) * 2u;
    const bool xw = wave < TF_NWX;
    const int LW = xw ? TF_NWI + 1 : TF_NWI;                    // W loads per lane per stage (wave-uniform)
    auto w_issue = [&](const int t) {                           // stage t = (k-step t / 3, tap t % 3) into ring slot t % TF_S
        const int ks = t / 3, tap = t - 3 * ks;
        unsigned char* wbase = smem + (t & (TF_S - 1)) * TF_WBYTES + wave * 1024;
        const uint32_t so = (uint32_t)(tap * C + ks * 32) * 2u;
#pragma unroll
        for (int j = 0; j < TF_NWI; ++j) VMV_BLDS16(w_rsrc, wbase + j * (TF_NW * 1024), wvo0 + (uint32_t)j * wstride, so);
        if (xw) VMV_BLDS16(w_rsrc, wbase + TF_NWI * (TF_NW * 1024), wvo0 + (uint32_t)TF_NWI * wstride, so);
    };

    // ---- prologue: tables, bias, zero padding rows, A stage 0, the W ring
    if constexpr (GN) {
        const float* gt = p.gn_table + (long)b * 2 * C;
        for (int i = tid; i < 2 * C; i += TF_NT) tab[i] = gt[i];
    }
    for (int i = tid; i < TF_BN; i += TF_NT) bias_lds[i] = (p.bias && n0 + i < p.N) ? p.bias[n0 + i] : 0.f;
    {   // rows [0, TF_PAD) and [TF_PAD + 192, 224) of both A buffers: the frames -1 / F (the loader never writes them)
        const u32x4_t z = u32x4_t{0u, 0u, 0u, 0u};
        for (int i = tid; i < 2 * 2 * TF_PAD * 8; i += TF_NT) {
            const int buf = i / (2 * TF_PAD * 8), rem = i - buf * (2 * TF_PAD * 8);
            const int row = rem >> 3, slot = rem & 7;
            const int rr = row < TF_PAD ? row : TF_PAD + TF_ROWS + (row - TF_PAD);
            *reinterpret_cast<u32x4_t*>(abuf + buf * TF_ABYTES + rr * 128 + slot * 16) = z;
        }
    }
    a_request(0);
    wait_vmcnt<0>();
    __syncthreads();                                            // tables visible (and my own A units landed)
    a_commit(0, 0); a_commit(0, 1); a_commit(0, 2);
    w_issue(0); w_issue(1); w_issue(2);
    if (NA > 1) a_request(1);                                   // (sits between W(2) and W(3) in the in-order queue: see the wait counts)
    w_issue(3);

    // ---- MFMA side
    f32x4_t acc[TF_WN][TF_WM];
#pragma unroll
    for (int j = 0; j < TF_WN; ++j)
#pragma unroll
        for (int i = 0; i < TF_WM; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fgrp = lane >> 4;
    const int fslot = fgrp ^ ((0x78 >> (2 * ((frow >> 2) & 3))) & 3);        // W fragments: physical slot of this lane's k-slice
    // B fragments of tap index tp (dt = tp - 1): lane row rr = wave_m * 48 + 16 i + frow + TF_PAD + dt * PT; (rr >> 1) & 7 is the
    // same for every i (16-row steps), so per tap: a lane-constant byte offset + a 64-byte half select
    uint32_t boff[3];                                            // (bit 6 = the 64-byte half that holds k-slots 0-3 of an even k-step for this lane's rows)
#pragma unroll
    for (int tp = 0; tp < 3; ++tp) {
        const int rr0 = wave_m * 16 * TF_WM + frow + TF_PAD + (tp - 1) * PT;
        const int sw = (rr0 >> 1) & 7;
        boff[tp] = (uint32_t)(rr0 * 128 + ((fgrp ^ (sw & 3)) << 4)) | ((uint32_t)(sw >> 2) << 6);
    }
    elem8_t bfr[2][TF_WM], wfr[2][TF_WH];
    auto read_b = [&](const int t, auto par_tag) {              // the 3 row-shifted B fragments of stage t
        constexpr int par = decltype(par_tag)::value;
        const int ks = t / 3, tap = t - 3 * ks;
        const unsigned char* ap = abuf + ((ks >> 1) & 1) * TF_ABYTES + (boff[tap] ^ ((uint32_t)(ks & 1) << 6));
#pragma unroll
        for (int i = 0; i < TF_WM; ++i) bfr[par][i] = __builtin_bit_cast(elem8_t, *reinterpret_cast<const u32x4_t*>(ap + i * 2048));
    };
    auto read_w = [&](const int t, auto h_tag) {                // the 5 W fragments of column half h of stage t
        constexpr int h = decltype(h_tag)::value;
        const u32x4_t* wp = reinterpret_cast<const u32x4_t*>(smem + (t & (TF_S - 1)) * TF_WBYTES) + (wave_n * 16 * TF_WN + h * 16 * TF_WH + frow) * 4 + fslot;
#pragma unroll
        for (int j = 0; j < TF_WH; ++j) wfr[h][j] = __builtin_bit_cast(elem8_t, wp[j * 16 * 4]);
    };
    auto mma = [&](auto par_tag, auto h_tag) {
        constexpr int par = decltype(par_tag)::value, h = decltype(h_tag)::value;
#pragma unroll
        for (int j = 0; j < TF_WH; ++j)
#pragma unroll
            for (int i = 0; i < TF_WM; ++i) acc[h * TF_WH + j][i] = VMV_MFMA16(wfr[h][j], bfr[par][i], acc[h * TF_WH + j][i], 0, 0, 0);
    };

    // W(0) landed (mine): W(1), W(2), [A(1)], W(3) may stay in flight
    wait_vmcnt_rt(3 * LW + (NA > 1 ? 3 : 0));
    __builtin_amdgcn_s_waitcnt(0xc07f);                         // my A stage 0 / padding writes are done
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_b(0, std::integral_constant<int, 0>{});
    read_w(0, std::integral_constant<int, 0>{});

    // One phase = one W stage = two half-phases of 15 MFMAs.  The loop body is TWO phases (the fragment-buffer parity is a compile-time
    // value; a body of the six phases of an A stage, with every decision static, costs ~50 more registers and spills), everything else
    // is wave-uniform run-time state (the A path is LDS-DMA: no register-returning load whose compiler-managed vmcnt a branch degrades).
    // The norm of stage a + 1 (GN): its six half-units (4 channels of one row each, ~30 VALU operations) ride in the six half-phases
    // between the wait of phase 6 a + 2 that covers their DMA and the barrier of phase 6 a + 5 that publishes them — (2, H1) -> 0,
    // (3, H0) -> 1, (3, H1) -> 2, (4, H0) -> 3, (4, H1) -> 4, (5, H0) -> 5 — and the two waves of a SIMD (w, w + 4) take them on
    // OPPOSITE sides of the half-phase's 15 MFMAs: waves 0-3 normalise first and multiply second, waves 4-7 multiply first, so one
    // wave's VALU work runs under the other's MFMAs (both normalising at once leaves the matrix pipe idle: +22-30 us per launch).
    const int T = 6 * NA;
    const bool late = wave >= TF_NW / 2;                         // this wave normalises AFTER the MFMAs of a half-phase
    auto phase = [&](const int t, const int t6, auto par_tag) {
        constexpr int par = decltype(par_tag)::value;
        using P0 = std::integral_constant<int, par>;
        using P1 = std::integral_constant<int, par ^ 1>;
        using H0 = std::integral_constant<int, 0>;
        using H1 = std::integral_constant<int, 1>;
        const int a_next = t / 6 + 1;
        read_w(t, H1{});
        __builtin_amdgcn_sched_barrier(0);
        const bool c0 = GN && t6 >= 3 && a_next < NA, c1 = GN && t6 >= 2 && t6 <= 4 && a_next < NA;
        if constexpr (GN) { if (c0 && !late) a_commit_half(a_next, 2 * (t6 - 3) + 1); }
        __builtin_amdgcn_sched_barrier(0);
        mma(P0{}, H0{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (GN) { if (c0 && late) a_commit_half(a_next, 2 * (t6 - 3) + 1); }
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < T) {
            // stage t + 1 landed for every wave, every wave done with ring slot t % S and with its A-buffer reads of phase t;
            // in flight behind W(t + 1): the later W stages already issued and, in phases 0 / 1, the A units of the next stage
            const int issued = t + TF_S < T ? t + TF_S : T;
            int allow = (issued - t - 2) * LW;
            if (t6 <= 1 && a_next < NA) allow += 3;
            // (allow = {0, 1, 2} x LW (+ 3), LW = 2 / 3: seven literals; a smaller literal only waits for more)
            if (allow >= 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else if (allow >= 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else if (allow >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (allow >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (allow >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (allow >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            read_b(t + 1, P1{});
            read_w(t + 1, H0{});
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (GN) { if (c1 && !late) a_commit_half(a_next, 2 * (t6 - 2)); }
        __builtin_amdgcn_sched_barrier(0);
        mma(P0{}, H1{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (GN) { if (c1 && late) a_commit_half(a_next, 2 * (t6 - 2)); }
        __builtin_amdgcn_sched_barrier(0);
        if (t6 == 5 && a_next + 1 < NA) a_request(a_next + 1);     // stage a + 2's units: in front of W(t + S) in the queue
        if (t + TF_S < T) w_issue(t + TF_S);                       // into the slot this phase's barrier freed
        __builtin_amdgcn_s_waitcnt(0xc07f);
    };
    int t6 = 0;
    for (int t = 0; t < T; t += 2) {
        phase(t, t6, std::integral_constant<int, 0>{});
        phase(t + 1, t6 + 1, std::integral_constant<int, 1>{});
        t6 = t6 == 4 ? 0 : t6 + 2;
    }

    // ---- epilogue: bias, activation -> LDS staging of whole rows -> 16-byte row stores (+ residual), as gemm_xglds.hip
    __syncthreads();                                            // ring / A buffers no longer read by anyone
#pragma unroll
    for (int j = 0; j < TF_WN; ++j) {
        const int nl = wave_n * 16 * TF_WN + 16 * j + 4 * fgrp;
        const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(bias_lds + nl);
#pragma unroll
        for (int i = 0; i < TF_WM; ++i) {
            f32x4_t v = acc[j][i] + bv;
            act_apply(v, p.act);
            u32x2_t o;
            o.x = pack_elem2(v.x, v.y); o.y = pack_elem2(v.z, v.w);
            *reinterpret_cast<u32x2_t*>(smem + (wave_m * 16 * TF_WM + 16 * i + frow) * TF_ROWB + nl * 2) = o;
        }
    }
    __syncthreads();
    constexpr int U = TF_BN >> 3;                               // 16-byte units per tile row
    uint16_t* outp = reinterpret_cast<uint16_t*>(p.out);
    const uint16_t* resp = reinterpret_cast<const uint16_t*>(p.residual);
    const float rs = p.res_scale != 0.f ? p.res_scale : 1.f;
    u32x4_t sd_prev = u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll 1
    for (int idx = tid; idx < RV * U; idx += TF_NT) {
        const int r = idx / U, u = idx - r * U;
        const int f = r / PT, j = r - f * PT;
        if (px0 + j >= P) continue;
        const long m = mb + (long)f * P + j;
        const int n = n0 + u * 8;
        if (n >= p.N) continue;
        u32x4_t v = *reinterpret_cast<const u32x4_t*>(smem + r * TF_ROWB + u * 16);
        if (resp) {
            const u32x4_t rr = *reinterpret_cast<const u32x4_t*>(resp + m * p.ldr + n);
            float a[8], c[8];
            unpack8(v, a); unpack8(rr, c);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += rs * c[e];
            v = pack8(a);
        }
        // store-data discipline (gemm_pglds.hip): the stored registers are a VALU-written copy, kept alive past the next LDS read
        __builtin_amdgcn_s_waitcnt(0xc07f);
        asm volatile("" ::"v"(sd_prev));
        u32x4_t sd;
        asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                     : "=&v"(sd.x), "=&v"(sd.y), "=&v"(sd.z), "=&v"(sd.w)
                     : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
        *reinterpret_cast<u32x4_t*>(outp + m * p.ldo + n) = sd;
        sd_prev = sd;
    }
}

int tfr_policy() {
    // VMV_GEMM_TFR (A/B experiments): 1 (default) = this kernel takes the eligible temporal convolutions, 0 = off
    static int pol = -1;
    if (pol < 0) { const char* e = getenv("VMV_GEMM_TFR"); pol = e ? atoi(e) : 1; }
    return pol;
}

}  // namespace

// host logic: can the frame-resident kernel serve *p at all (forced tile or policy)?
bool vmv_gemm_tfr_supported(const VmvGemmParams& p) {
    if (p.nseg != 3 || p.F < 12 || p.F > 24 || p.P <= 0) return false;
    const VmvGemmSeg& s0 = p.seg[0];
    for (int i = 0; i < 3; ++i) {
        const VmvGemmSeg& sg = p.seg[i];
        if (sg.mode != VMV_SEG_TEMPORAL || sg.d0 != i - 1 || sg.src != s0.src || sg.ld != s0.ld || sg.k != s0.k) return false;
    }
    const int C = s0.k;
    if ((C & 63) || C > TF_MAXC || p.ktot != 3 * C || (p.N % TF_BN)) return false;
    if ((long)p.M % ((long)p.F * p.P)) return false;
    if (p.epilogue != VMV_EPI_NONE || p.rowvec || p.rowstat || p.colsum || p.ln_eps > 0.f || p.wgroup_rows || p.ksplit > 1 || p.out_fp32) return false;
    if ((p.ldo & 7) || !vmv_aligned16(p.out) || (p.residual && ((p.ldr & 7) || !vmv_aligned16(p.residual)))) return false;
    if (p.gn_table && (p.gn_rows_per_stat != p.F * p.P || !vmv_aligned16(p.gn_table))) return false;
    if ((long)p.M * s0.ld * 2 >= (1L << 31) - 65536 || (long)(p.N + TF_BN) * p.ktot * 2 >= (1L << 31) - 65536) return false;
    return true;
}

// policy: taken when its tiles (samples x pixel groups x N / 320) fill WHOLE rounds of the 256 CUs — >= 200 tiles and >= 95 % of the
// last round.  Measured (tools/experiments/tfr_bench.py, profiles/r5_tfr_bench.log): a tile runs at the same ~3.4 TFLOP/s per CU as the
// 256 x 320 tile of gemm_xglds.hip, so the frame-resident form wins where its grid quantises better — the first level at 24 x 32 x 32:
// 256 tiles = one round, 36-39 us against 43 us, folded norm 62-67 us against 74 us for statistics + apply + convolution — and loses
// where it does not: 24 x 40 x 64 has 640 tiles = 2.5 rounds (100-111 us against 90-94 us).
bool vmv_gemm_tfr_preferred(const VmvGemmParams& p) {
    if (!tfr_policy() || !vmv_gemm_tfr_supported(p)) return false;
    const int PT = TF_ROWS / p.F;
    const long tiles = (long)(p.M / ((long)p.F * p.P)) * ((p.P + PT - 1) / PT) * (p.N / TF_BN);
    const long rounds = (tiles + 255) / 256;
    static double fill_min = -1.0;
    if (fill_min < 0.0) { const char* e = getenv("VMV_TFR_FILL"); fill_min = e ? atof(e) : 0.95; }      // (A/B experiments)
    return tiles >= 200 && (double)tiles / (double)(rounds * 256) >= fill_min;
}

int vmv_gemm_tfr_launch(const VmvGemmParams& p, hipStream_t st) {
    if (!vmv_gemm_tfr_supported(p)) return VMV_GLDS_UNSUPPORTED;
    const int PT = TF_ROWS / p.F;
    const int nb = (int)(p.M / ((long)p.F * p.P));
    const int tiles_g = (p.P + PT - 1) / PT, tiles_n = p.N / TF_BN;
    const dim3 grid((unsigned)(nb * tiles_g * tiles_n));
    if (p.gn_table) {
        static std::atomic<unsigned long long> attr_gn{0};
        if (const int rc = vmv_lds_attr_once(attr_gn, reinterpret_cast<const void*>(&gemm_tfr_kernel<true>), TF_LDS)) return rc;
        VMV_LAUNCH((gemm_tfr_kernel<true>), grid, dim3(TF_NT), TF_LDS, st, p, tiles_g, tiles_n, PT);
    } else {
        static std::atomic<unsigned long long> attr_pl{0};
        if (const int rc = vmv_lds_attr_once(attr_pl, reinterpret_cast<const void*>(&gemm_tfr_kernel<false>), TF_LDS)) return rc;
        VMV_LAUNCH((gemm_tfr_kernel<false>), grid, dim3(TF_NT), TF_LDS, st, p, tiles_g, tiles_n, PT);
    }
    return vmv_launch_status();
}

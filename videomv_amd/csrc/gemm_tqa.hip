// gemm_tqa.hip — q | k | v projection AND the per-pixel temporal attention in ONE launch (VMV_EPI_TATTN, vmv.h): the first half of a
// TemporalTransformer's attention (tools/modules/unet/util.py:1043-1089 -> BasicTransformerBlock :536-540 ->
// MemoryEfficientCrossAttention :230-268 with context = None: to_q / to_k / to_v -> softmax(q k^T / sqrt(64)) v over the F frames of a
// pixel).  Round 6, VERDICT r5 item 2.
//
// Why.  In the two-kernel form the row-stationary GEMM writes q | k | v of every (frame, pixel) row — 236 MB at the first level of the
// 24 x 40 x 64 plan — and attn_short_kernel reads them straight back to do 3.8 GFLOP on them at 45 TFLOP/s (81-84 us, HBM-bound); both
// launches move 3-4 x the bytes of the activation they started from.  Here a wave keeps ALL F frames of 48 / F pixels — 48 rows = three
// 16-row MFMA fragments, the whole K = 320 range — in registers (gemm_rs.hip's row-stationary scheme), streams the weight matrix
// through the same three-stage LDS ring in head-major order [head][q | k | v][64 rows], and finishes a head's attention in registers
// before the next head's weights arrive: q, k and v never exist in memory, only the 64-column attention output of each head is stored.
//
// No lane ever needs another lane's data between the projection and the attention, because the contraction indices can be permuted:
//   * k and q come out of the TRANSPOSED product (MFMA A = W fragment, B = activation fragment): lane (u, g) holds channels 4 g .. 4 g + 3
//     of row u for every 16-column tile.  Two tiles make 8 values per lane = "k-slice g" of a 32-deep MFMA step — a permutation of the 32
//     head channels, the SAME one for q and k, so S^T = K Q^T over those slices is the exact dot product:
//         s[ik][iq] = MFMA(A = k fragment ik, B = q fragment iq)  ->  lane (u, g): S[query = row u of iq][keys = rows 4 g + r of ik];
//   * v comes out of the PLAIN product (A = activation fragment, B = W fragment: the two operands have the same lane layout, so swapping
//     them transposes the result): lane (u, g) holds V[rows 4 g + r of fragment i][channel u] — which IS the A operand V^T[d = u][keys
//     4 g + r] of a 16-deep MFMA whose B operand P^T[keys 4 g + r][query u] are the exponentiated S registers above:
//         o[t] += MFMA16(A = v fragment (ik, tile t), B = p fragment ik)  ->  lane (u, g): O[query u][channels 16 t + 4 g + r],
//     the layout gemm_rs.hip's store path (two v_permlane16_swap -> 16-byte stores) starts from.
// Rows of a wave: tile row tr = 16 i + u -> pixel tr / F, frame tr % F (F = 24: fragment 1 holds frames 16-23 of the first pixel and
// 0-7 of the second; scores between different pixels are masked, tile pairs that share no pixel are skipped).
// Per head and wave: 360 projection MFMAs + 14 (S) + 28 (P V; 16-deep) and ~250 VALU operations of softmax, no LDS traffic but W.
#include "gemm_glds_common.h"
#include <cstdlib>
#include <type_traits>

using namespace vmvg;

namespace {

constexpr int TQ_RT = 3, TQ_KS = 10, TQ_K = 320;
constexpr int TQ_NW = 8, TQ_NT = 512;
constexpr int TQ_ROWS = 16 * TQ_RT;                         // tile rows of a wave (F x pixels)
constexpr int TQ_RB = TQ_K * 2;                             // bytes per W row
constexpr int TQ_SPR = TQ_RB / 16;                          // 16-byte slots per W row
constexpr int TQ_CHUNK = 40960, TQ_CROWS = TQ_CHUNK / TQ_RB;      // a chunk = 64 W rows = one of k / v / q of one head
constexpr int TQ_STAGES = 3;
constexpr int TQ_P = TQ_CHUNK / 1024 / TQ_NW;               // LDS-DMA wave-instructions per wave per chunk (5)
constexpr int TQ_MAXCOLS = 3840;                            // 20 heads
constexpr int TQ_LDS = TQ_STAGES * TQ_CHUNK + TQ_MAXCOLS * 4;
static_assert(TQ_CROWS == 64 && TQ_P * TQ_NW * 1024 == TQ_CHUNK && TQ_LDS <= 160 * 1024, "chunk geometry");
constexpr float TQ_NEG = -1.0e30f;

#if defined(VMV_BUILD_BF16)
typedef __attribute__((ext_vector_type(4))) short tq_elem4_t;
#define TQ_MFMA_K16 __builtin_amdgcn_mfma_f32_16x16x16bf16_1k
#else
typedef __attribute__((ext_vector_type(4))) _Float16 tq_elem4_t;
#define TQ_MFMA_K16 __builtin_amdgcn_mfma_f32_16x16x16f16
#endif

VMV_DEV u32x4_t tq_swap16(u32x4_t v) {          // gemm_rs.hip swap16_xz_yw (see there for the wait states)
    uint32_t x = v.x, y = v.y, z = v.z, w = v.w;
    asm("s_nop 3\n\tv_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3\n\ts_nop 1" : "+v"(x), "+v"(y), "+v"(z), "+v"(w));
    return u32x4_t{x, y, z, w};
}

// NEED: bit 3 iq + ik set = query fragment iq and key fragment ik share a pixel (compile-time: the unneeded score tiles, their P^T
// registers and MFMAs do not exist).  F | 16: the diagonal; F = 24 / 12 / 6 / 3: the band; F = 48: all nine.
constexpr uint32_t tq_need_mask(int F) {
    uint32_t need = 0;
    for (int iq = 0; iq < TQ_RT; ++iq)
        for (int ik = 0; ik < TQ_RT; ++ik) {
            const int q0 = (16 * iq) / F, q1 = (16 * iq + 15) / F, k0 = (16 * ik) / F, k1 = (16 * ik + 15) / F;
            if (q0 <= k1 && k0 <= q1) need |= 1u << (3 * iq + ik);
        }
    return need;
}
constexpr uint32_t TQ_DIAG = tq_need_mask(16), TQ_BAND = tq_need_mask(24), TQ_FULL = tq_need_mask(48);
static_assert(TQ_DIAG == 0x111u && TQ_BAND == 0x1bbu && TQ_FULL == 0x1ffu && tq_need_mask(12) == TQ_BAND && tq_need_mask(6) == TQ_BAND &&
              tq_need_mask(3) == TQ_BAND && tq_need_mask(8) == TQ_DIAG && tq_need_mask(1) == TQ_DIAG, "pixel / fragment overlap patterns");

template <bool LN, uint32_t NEED>
__global__ __launch_bounds__(512, 1) void gemm_tqa_kernel(const VmvGemmParams p, const int ntiles, const int heads, const int npix) {
    VMV_KERNEL_ENTER();
    constexpr int RT = TQ_RT, KS = TQ_KS, RB = TQ_RB, P = TQ_P;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fgrp = lane >> 4;
    const int F = p.F, PX = p.P;
    const int PPW = TQ_ROWS / F;                                // pixels per wave

    // ---- persistent (row tile, head) items (round 6, second form).  One block per CU; the items — ntiles x heads, tile-major — are dealt
    //      out in contiguous, equally long ranges (lengths differ by at most one item), so 320 row tiles on 256 CUs cost 7 items per CU
    //      instead of two whole tiles = 10 (24 x 40 x 64), and 128 row tiles cost 3 instead of 5 with half the chip idle (24 x 32 x 32).
    //      A block re-loads its rows only when its range crosses a tile boundary (at most twice more); W keeps streaming through the
    //      ring across items: chunk c of the block is section c % 3 (q, k, v) of head (i0 + c / 3) % heads.
    const int nblk = gridDim.x;
    int logical;
    {
        const int bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const long nitems = (long)ntiles * heads;
    const int i0 = (int)(((long)logical * nitems) / nblk), i1 = (int)(((long)(logical + 1) * nitems) / nblk);
    const int nit = i1 - i0;                                    // >= 1 (the grid never exceeds the item count)

    // ---- the wave's 48 tile rows: tile row tr = 16 i + frow -> pixel gp0 + tr / F, frame tr % F -> global row (b F + f) P + pp
    const VmvGemmSeg& sg = p.seg[0];
    const int lanecol = (fgrp & 1) * 16 + (fgrp >> 1) * 8;      // after the lane swaps: this lane's 8 consecutive columns of a 32-column pair
    uint32_t ovo[RT];
    int qlo[RT];                                                // first tile row of the pixel that owns tile row 16 i + frow
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(sg.src), 0, (uint32_t)p.M * (uint32_t)sg.ld * 2u, SRD_FLAGS);
    u32x4_t a[RT][KS];
    auto load_tile = [&](const int tile) __attribute__((always_inline)) {                      // the rows of row tile `tile` into the registers (rows outside the tensor: zeros)
        const int gp0 = (tile * TQ_NW + wave) * PPW;            // first (sample, pixel) index of this wave
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int tr = 16 * i + frow;
            const int px = tr / F, f = tr - px * F;
            const int gp = gp0 + px;
            const bool ok = gp < npix;
            const int b = gp / PX, pp = gp - b * PX;
            const long m = ((long)b * F + f) * PX + pp;
            const uint32_t avo = ok ? (uint32_t)((m * sg.ld + 8 * fgrp) * 2) : OOB;
            ovo[i] = ok ? (uint32_t)((m * p.ldo + lanecol) * 2) : OOB;
            qlo[i] = px * F;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) a[i][kk] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, avo + (uint32_t)(kk * 64), 0, 0);
        }
    };
    int cur_tile = i0 / heads;
    load_tile(cur_tile);

    // ---- bias strip (the folded LayerNorm's W beta, head-major like W), then the W ring (gemm_rs.hip: swizzle (r >> 1) & 7 on the SOURCE)
    float* bias_lds = reinterpret_cast<float*>(smem + TQ_STAGES * TQ_CHUNK);
    {
        const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias ? (uint32_t)p.N * 4u : 0u, SRD_FLAGS);
        for (int q = wave; q * 256 < p.N; q += TQ_NW)
            VMV_BLDS16(b_rsrc, reinterpret_cast<unsigned char*>(bias_lds) + q * 1024, (uint32_t)(q * 256 + 4 * lane) * 4u, 0);
    }
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, (uint32_t)p.N * (uint32_t)p.ktot * 2u, SRD_FLAGS);
    auto issue_chunk = [&](int c, int slot) {
        unsigned char* base = smem + slot * TQ_CHUNK + wave * (P * 1024);
        const int it = c / 3, hc = (i0 + it) % heads;          // chunk c = section c % 3 of the head of item i0 + c / 3
        const uint32_t so = (uint32_t)((3 * hc + (c - 3 * it)) * TQ_CROWS * p.ktot) * 2u;
        int ln = lane;
        asm volatile("" : "+v"(ln));                           // (recomputed per chunk instead of five more live registers: gemm_rs.hip)
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const int u = wave * (P * 64) + q * 64 + ln;
            const int r = u / TQ_SPR, s = u - r * TQ_SPR;
            const int sw = (r >> 1) & 7;
            VMV_BLDS16(w_rsrc, base + q * 1024, (uint32_t)(r * p.ktot + (s ^ sw) * 8) * 2u, so);
        }
    };
    const int NC = 3 * nit;
    for (int c = 0; c < TQ_STAGES; ++c) issue_chunk(c, c);       // (NC >= 3 always)

    // ---- LayerNorm of the resident rows (two-pass, fp32; gemm_rs.hip): the plain product with W' = W diag(gamma) follows
    auto layernorm_rows = [&]() __attribute__((always_inline)) {        // (called twice: out of line the rows would live in scratch)
    if constexpr (LN) {
        const float inv_k = 1.0f / (float)TQ_K;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            float s1 = 0.f;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                elem_dot2c(s1, a[i][kk].x, VMV_ELEM_ONE2); elem_dot2c(s1, a[i][kk].y, VMV_ELEM_ONE2);
                elem_dot2c(s1, a[i][kk].z, VMV_ELEM_ONE2); elem_dot2c(s1, a[i][kk].w, VMV_ELEM_ONE2);
            }
            asm volatile("s_nop 4" : "+v"(s1));                // (DOT-pipe result -> VALU read: gemm_rs.hip)
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
    };
    layernorm_rows();

    constexpr uint32_t need = NEED;
    const float sc = p.epi_scale * 1.44269504088896341f;       // exp2 domain

    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, SRD_RECORDS, SRD_FLAGS);

    // ---- first chunk (and the bias strip, issued before it) visible to every wave
    wait_vmcnt_rt(2 * P);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    const int fsw = (frow >> 1) & 7;
    int foff[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) foff[r] = frow * RB + (fgrp ^ (fsw & 3)) * 16 + ((r ^ (fsw >> 2)) - r) * 64;
    // one pair of 16-row W tiles against the resident rows.  SWAP = false: transposed product (c: row frow, channels 4 fgrp + r of the
    // tile); SWAP = true: plain product (c: rows 4 fgrp + r, channel frow of the tile).  W fragments one k-step ahead (gemm_rs.hip).
    auto mma_pair = [&](const unsigned char* sbase, const int q, f32x4_t (&c0)[RT], f32x4_t (&c1)[RT], auto swap_tag) {
        constexpr bool SWAP = decltype(swap_tag)::value;
        const unsigned char* tb[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) tb[r] = sbase + 32 * q * RB + foff[r];
#pragma unroll
        for (int i = 0; i < RT; ++i) { c0[i] = f32x4_t{0.f, 0.f, 0.f, 0.f}; c1[i] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
        u32x4_t w0[2], w1[2];
        auto rd = [&](const int kk, u32x4_t& x0, u32x4_t& x1) {
            const unsigned char* t = tb[kk & 1] + 64 * kk;
            x0 = *reinterpret_cast<const u32x4_t*>(t);
            x1 = *reinterpret_cast<const u32x4_t*>(t + 16 * RB);
        };
        rd(0, w0[0], w1[0]);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int cur = kk & 1;
            if (kk + 1 < KS) rd(kk + 1, w0[cur ^ 1], w1[cur ^ 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                if constexpr (SWAP) {
                    c0[i] = VMV_MFMA16(__builtin_bit_cast(elem8_t, a[i][kk]), __builtin_bit_cast(elem8_t, w0[cur]), c0[i], 0, 0, 0);
                    c1[i] = VMV_MFMA16(__builtin_bit_cast(elem8_t, a[i][kk]), __builtin_bit_cast(elem8_t, w1[cur]), c1[i], 0, 0, 0);
                } else {
                    c0[i] = VMV_MFMA16(__builtin_bit_cast(elem8_t, w0[cur]), __builtin_bit_cast(elem8_t, a[i][kk]), c0[i], 0, 0, 0);
                    c1[i] = VMV_MFMA16(__builtin_bit_cast(elem8_t, w1[cur]), __builtin_bit_cast(elem8_t, a[i][kk]), c1[i], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using T_ = std::integral_constant<bool, true>;
    using F_ = std::integral_constant<bool, false>;

    int c = 0, slot = 0;
    // chunk c consumed: chunk c + 1 landed for every wave, slot of chunk c refilled with chunk c + 3.  `extra` = this wave's stores that
    // were issued behind chunk c + 1's DMA (the previous and the current chunk's): they and chunk c + 2's DMA may stay in flight.
    auto chunk_end = [&](const int extra) {
        if (c + 1 < NC) {
            wait_vmcnt_rt((c + 2 < NC ? P : 0) + extra);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (c + TQ_STAGES < NC) issue_chunk(c + TQ_STAGES, slot);
        }
        ++c;
        slot = slot + 1 == TQ_STAGES ? 0 : slot + 1;
    };
    // a fragment of 8 (k, q) or 4 (v) 16-bit values from fp32 accumulators
    auto pack_kq = [&](const f32x4_t& v0, const f32x4_t& v1) {
        return u32x4_t{pack_elem2(v0.x, v0.y), pack_elem2(v0.z, v0.w), pack_elem2(v1.x, v1.y), pack_elem2(v1.z, v1.w)};
    };

    for (int it = 0; it < nit; ++it) {
        const int tile = (i0 + it) / heads, h = (i0 + it) - tile * heads;
        if (tile != cur_tile) {         // the range crossed into the next row tile: new rows (every MFMA on the old ones has issued; the
            cur_tile = tile;            // compiler's own waits cover the loads; the ring's DMAs in flight are untouched)
            load_tile(tile);
            layernorm_rows();
        }
        // Chunk order q, k, v.  After k: S^T, mask, softmax -> the P^T fragments (14-18 registers) and 1 / l replace q and k (48); the v
        // chunk then multiplies each pair of 16-channel v tiles into the output as soon as the pair is projected.  (k, v, q with the
        // whole attention at the end keeps q, k AND v alive next to the projection's accumulators: 119 spilled registers.)
        u32x4_t qf[RT][2], kf[RT][2];                           // [row fragment][32-channel half]
        {   // ---- chunk 3 h: q rows of W
            const unsigned char* sbase = smem + slot * TQ_CHUNK;
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                f32x4_t c0[RT], c1[RT];
                mma_pair(sbase, pr, c0, c1, F_{});
                const int nrel = 192 * h + 32 * pr;
                const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(bias_lds + nrel + 4 * fgrp);
                const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(bias_lds + nrel + 16 + 4 * fgrp);
#pragma unroll
                for (int i = 0; i < RT; ++i) qf[i][pr] = pack_kq(c0[i] + b0, c1[i] + b1);
            }
            chunk_end(it > 0 ? 2 * RT : 0);
        }
        {   // ---- chunk 3 h + 1: k rows of W
            const unsigned char* sbase = smem + slot * TQ_CHUNK;
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                f32x4_t c0[RT], c1[RT];
                mma_pair(sbase, pr, c0, c1, F_{});
                const int nrel = 192 * h + 64 + 32 * pr;
                const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(bias_lds + nrel + 4 * fgrp);
                const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(bias_lds + nrel + 16 + 4 * fgrp);
#pragma unroll
                for (int i = 0; i < RT; ++i) kf[i][pr] = pack_kq(c0[i] + b0, c1[i] + b1);
            }
        }
        // ---- S^T tiles of every query fragment, masked to the query's own pixel; softmax over its F keys
        u32x2_t pf[RT][RT];                                     // P^T fragments [query fragment][key fragment] (16-bit, unnormalised)
        float linv[RT];
#pragma unroll
        for (int iq = 0; iq < RT; ++iq) {
            f32x4_t s[RT];
            const int kb = 4 * fgrp - qlo[iq];
            float mx = TQ_NEG;
#pragma unroll
            for (int ik = 0; ik < RT; ++ik) {
                if ((need >> (3 * iq + ik)) & 1u) {
                    s[ik] = VMV_MFMA16(__builtin_bit_cast(elem8_t, kf[ik][0]), __builtin_bit_cast(elem8_t, qf[iq][0]), (f32x4_t{0.f, 0.f, 0.f, 0.f}), 0, 0, 0);
                    s[ik] = VMV_MFMA16(__builtin_bit_cast(elem8_t, kf[ik][1]), __builtin_bit_cast(elem8_t, qf[iq][1]), s[ik], 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if ((unsigned)(kb + 16 * ik + r) >= (unsigned)F) s[ik][r] = TQ_NEG;
                    mx = vmax3(mx, s[ik][0], s[ik][1]);
                    mx = vmax3(mx, s[ik][2], s[ik][3]);
                }
            }
            mx = xor32_max3(xor16_max(mx), TQ_NEG);
            const float nm = -mx * sc;
            float psum = 0.f;
#pragma unroll
            for (int ik = 0; ik < RT; ++ik) {
                pf[iq][ik] = u32x2_t{0u, 0u};
                if ((need >> (3 * iq + ik)) & 1u) {
                    const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[ik][0], sc, nm)), e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[ik][1], sc, nm));
                    const float e2 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[ik][2], sc, nm)), e3 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[ik][3], sc, nm));
                    psum += (e0 + e1) + (e2 + e3);
                    pf[iq][ik] = u32x2_t{pack_elem2(e0, e1), pack_elem2(e2, e3)};
                }
            }
            linv[iq] = 1.0f / xor16_32_sum(psum);
            __builtin_amdgcn_sched_barrier(0);                  // one query fragment at a time (three interleaved: 36 more live registers)
        }
        chunk_end(0);
        {   // ---- chunk 3 h + 2: v rows of W (plain product), each pair of 16-channel tiles straight into O^T = V^T P^T and out
            const unsigned char* sbase = smem + slot * TQ_CHUNK;
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                f32x4_t c0[RT], c1[RT];
                mma_pair(sbase, pr, c0, c1, T_{});
                const int nrel = 192 * h + 128 + 32 * pr;
                const float b0 = bias_lds[nrel + frow], b1 = bias_lds[nrel + 16 + frow];
                u32x2_t v0[RT], v1[RT];                         // v fragments of the two tiles: [key fragment]
#pragma unroll
                for (int i = 0; i < RT; ++i) {
                    v0[i] = u32x2_t{pack_elem2(c0[i].x + b0, c0[i].y + b0), pack_elem2(c0[i].z + b0, c0[i].w + b0)};
                    v1[i] = u32x2_t{pack_elem2(c1[i].x + b1, c1[i].y + b1), pack_elem2(c1[i].z + b1, c1[i].w + b1)};
                }
#pragma unroll
                for (int iq = 0; iq < RT; ++iq) {
                    f32x4_t o0 = f32x4_t{0.f, 0.f, 0.f, 0.f}, o1 = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ik = 0; ik < RT; ++ik) {
                        if ((need >> (3 * iq + ik)) & 1u) {
                            o0 = TQ_MFMA_K16(__builtin_bit_cast(tq_elem4_t, v0[ik]), __builtin_bit_cast(tq_elem4_t, pf[iq][ik]), o0, 0, 0, 0);
                            o1 = TQ_MFMA_K16(__builtin_bit_cast(tq_elem4_t, v1[ik]), __builtin_bit_cast(tq_elem4_t, pf[iq][ik]), o1, 0, 0, 0);
                        }
                    }
                    o0 *= linv[iq]; o1 *= linv[iq];
                    u32x4_t ov = tq_swap16(pack_kq(o0, o1));
                    __builtin_amdgcn_raw_buffer_store_b128(ov, out_rsrc, ovo[iq], (uint32_t)(64 * h + 32 * pr) * 2u, 0);
                    asm volatile("s_nop 7" ::"v"(ov.x), "v"(ov.y), "v"(ov.z), "v"(ov.w) : "memory");       // (store-data discipline: gemm_rs.hip)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            chunk_end(2 * RT);
        }
    }
}

int tqa_ncu() {          // CUs of whole XCDs (one block per CU; gemm_rs.hip ncu_whole_xcds)
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0, n = 0;
        if (vmv_dry_run || hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        if (n < 8) n = 8;
        if (vmv_dry_run) return n & ~7;      // (validation without a device must not cache the guess)
        ncu = n & ~7;
    }
    return ncu;
}

int tqa_policy() {
    // VMV_GEMM_TQA (A/B experiments): 1 (default) = the engine records the fused launch where it is supported, 0 = the two-kernel form
    static int pol = -1;
    if (pol < 0) { const char* e = getenv("VMV_GEMM_TQA"); pol = e ? atoi(e) : 1; }
    return pol;
}

}  // namespace

// host logic: can the fused kernel serve *p?  (VMV_EPI_TATTN has no other home: vmv_gemm returns VMV_EINVAL when this says no)
bool vmv_gemm_tqa_supported(const VmvGemmParams& p) {
    if (p.epilogue != VMV_EPI_TATTN) return false;
    if (p.nseg != 1 || p.seg[0].mode != VMV_SEG_LINEAR || p.seg[0].k != p.ktot || p.ktot != TQ_K) return false;
    if (p.F <= 0 || p.F > TQ_ROWS || (TQ_ROWS % p.F) || p.P <= 0 || ((long)p.M % ((long)p.F * p.P))) return false;
    if ((p.N % 192) || p.N > TQ_MAXCOLS || p.N < 192) return false;
    if (p.ksplit > 1 || p.out_fp32 || p.rowvec || p.rowstat || p.residual || p.act != VMV_ACT_NONE || p.wgroup_rows != 0 || p.gn_table || p.gn_silu) return false;
    if (p.colsum && !(p.ln_eps > 0.f)) return false;
    if (!(p.epi_scale > 0.f)) return false;
    if ((p.ldo & 7) || (p.ldo < p.N / 3) || !vmv_aligned16(p.out)) return false;
    if ((long)(p.M + 512) * p.seg[0].ld * 2 >= (1L << 31) - 65536 || (long)(p.M + 512) * p.ldo * 2 >= (1L << 31) - 65536) return false;
    if ((long)p.N * p.ktot * 2 >= (1L << 31) - 65536) return false;
    return true;
}

// policy: taken when the (row tile, head) items give most of the chip a block (persistent blocks, one item or more each).  Measured
// (tools/experiments/tqa_bench.py, profiles/r6_tqa_bench.log): 102 us against 167 us for the two launches at the first level of
// 24 x 40 x 64, 46 against 71 at 24 x 32 x 32, 24 against 54 on rank 0 of 8 (40 row tiles x 5 heads = 200 items).
bool vmv_gemm_tqa_preferred(const VmvGemmParams& p) {
    if (!tqa_policy() || !vmv_gemm_tqa_supported(p)) return false;
    const long npix = (long)(p.M / ((long)p.F * p.P)) * p.P;
    const long tiles = (npix + TQ_NW * (TQ_ROWS / p.F) - 1) / (TQ_NW * (TQ_ROWS / p.F));
    static long min_items = -1;
    if (min_items < 0) { const char* e = getenv("VMV_TQA_MIN_ITEMS"); min_items = e ? atol(e) : 128; }      // (tests / A/B experiments)
    return tiles * (p.N / 192) >= min_items;
}

int vmv_gemm_tqa_launch(const VmvGemmParams& p, hipStream_t st) {
    if (!vmv_gemm_tqa_supported(p)) return VMV_GLDS_UNSUPPORTED;
    const int npix = (int)((p.M / ((long)p.F * p.P)) * p.P);
    const int ppb = TQ_NW * (TQ_ROWS / p.F);
    const int ntiles = (npix + ppb - 1) / ppb;
    const int heads = p.N / 192;
    const long nitems = (long)ntiles * heads;
    const int nblk = (int)(nitems < tqa_ncu() ? nitems : tqa_ncu());
    const uint32_t need = tq_need_mask(p.F);
    const bool ln = p.colsum != nullptr;
#define TQ_LAUNCH(LNV, NEEDV)                                                                                                            \
    do {                                                                                                                                 \
        static std::atomic<unsigned long long> attr{0};                                                                                  \
        if (const int rc = vmv_lds_attr_once(attr, reinterpret_cast<const void*>(&gemm_tqa_kernel<LNV, NEEDV>), TQ_LDS)) return rc;      \
        VMV_LAUNCH((gemm_tqa_kernel<LNV, NEEDV>), dim3(nblk), dim3(TQ_NT), TQ_LDS, st, p, ntiles, heads, npix);                          \
    } while (0)
    if (need == TQ_BAND) { if (ln) TQ_LAUNCH(true, TQ_BAND); else TQ_LAUNCH(false, TQ_BAND); }
    else if (need == TQ_DIAG) { if (ln) TQ_LAUNCH(true, TQ_DIAG); else TQ_LAUNCH(false, TQ_DIAG); }
    else { if (ln) TQ_LAUNCH(true, TQ_FULL); else TQ_LAUNCH(false, TQ_FULL); }
#undef TQ_LAUNCH
    return vmv_launch_status();
}

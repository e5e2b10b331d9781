// gemm_ff.hip — the FeedForward of a transformer block in ONE launch at the UNet's largest level (C = 320):
//     out = x + W2 . ( (W1x LN(x) + b1x) * gelu(W1g LN(x) + b1g) ) + b2                  (util.py:541-573, :536-540)
// Same row-stationary idea as gemm_rs.hip, taken one GEMM further: the 4C-wide hidden activation never exists in memory.
//
// Why: after gemm_rs the two FF GEMMs of an L0 block were 214 + 165 us (profiles/r3_ops_40x64.tsv) and moved 865 MB, 630 MB of
// it the hidden tensor written by one launch and read back by the next (M = 122 880 rows x 1280 channels x 2 B, twice).  The
// down projection (K = 1280) is the slowest linear left (608 TFLOP/s): its A operand cannot be made resident and it is bound by
// HBM + the CU's LDS-DMA path.  Fused, the block reads x once and writes out once (236 MB) and W1 / W2 stream through LDS.
//
// Structure.  A wave owns 16 rows: LayerNorm-ed x in 40 registers (the B operand of every FF1 MFMA), the 16 x 320 output tile in
// 80 accumulator registers for the whole kernel.  The hidden dimension is walked in chunks of 32 channels:
//   FF1   the chunk's four 16-row W1 tiles (x | gate | x | gate: packing.geglu_interleave) x 10 k-steps -> 4 accumulators;
//   GELU  h = (x + b1x) * gelu(gate + b1g), rounded to the 16-bit element type exactly where the two-launch form rounds it;
//   FF2   the MFMA OUTPUT layout of two 16-channel tiles (lane = row, 4 consecutive channels each) IS a B-operand fragment of a
//         32-deep k-step once W2's K axis is stored in the matching order (packing.ff_down_permute: per 32 channels
//         [0-3, 16-19 | 4-7, 20-23 | 8-11, 24-27 | 12-15, 28-31]) — no shuffle, no LDS round trip: 20 MFMAs add the chunk to out.
// A block is 8 waves = 128 rows sharing the weight stream: per chunk 40 KB of W1 (64 rows x 640 B, gemm_rs's swizzle) + 20 KB of
// W2 (320 rows x 64 B, gemm_xglds's 4-entry slot swizzle) by LDS-DMA into a two-stage ring; one barrier per chunk (60 MFMAs per
// wave).
//
// Measured (M = 122 880, f16; tools/experiments/ff_bench.py and run_ff_ab.sh, one box): 383 us fused against 445 us for the two
// launches back to back in isolation, but inside a step the two launches take 214 + 165 us and the step is 51.61 ms with the
// fused kernel against 51.52 ms without — so the engine leaves it OFF (VMV_FF_FUSED=1 opts in) and the entry point is kept for
// callers whose memory, not time, is the constraint (no M x 4C hidden tensor).  Why it stops at 790 TFLOP/s: with one row tile
// per wave every MFMA needs its own 1-KB weight fragment, i.e. the eight waves ask the LDS read port for 512 B/clk while issuing
// MFMAs at peak, twice what it delivers, and the waves of a block move in lockstep (one barrier per 60 MFMAs), so the fragment
// reads, the GELU (VALU) and the DMA issue of the next chunk add instead of overlapping.  Ablations (wrong results, same box):
// no MFMAs 314 us, GELU -> identity 336 us, no barrier / DMA 308 us, deeper fragment prefetch (2 k-steps / 6 tiles, 3 / 8) 390 /
// 395 us.  The way forward is 32 x 32 x 16 MFMAs over 32 rows per wave PAIR with the output columns split between the two waves
// and the hidden chunk exchanged through LDS (half the fragment bytes per MFMA); 80 + 80 + 32 registers — not built.
#include "gemm_glds_common.h"

using namespace vmvg;

namespace {

constexpr int FF_C = 320, FF_KS = FF_C / 32, FF_NT = FF_C / 16;       // channels, k-steps of FF1, output tiles of FF2
constexpr int FF_HC = 32;                                                // hidden channels per chunk
constexpr int FF_W1_BYTES = 2 * FF_HC * FF_C * 2;                        // 64 rows x 640 B
constexpr int FF_W2_BYTES = FF_C * FF_HC * 2;                            // 320 rows x 64 B
constexpr int FF_STAGE = FF_W1_BYTES + FF_W2_BYTES;                      // 61 440
constexpr int FF_B1_BYTES = 8 * FF_C * 4, FF_B2_BYTES = FF_C * 4;
constexpr int FF_LDS = 2 * FF_STAGE + FF_B1_BYTES + FF_B2_BYTES;
static_assert(FF_LDS <= 160 * 1024, "LDS budget");

#ifndef VMV_FF_PF1
#define VMV_FF_PF1 1       // experiments: FF1 fragment prefetch distance in k-steps (4 fragments each)
#endif
#ifndef VMV_FF_PF2
#define VMV_FF_PF2 3       // experiments: FF2 fragment prefetch distance in output tiles
#endif
#ifndef VMV_FF_ABLATE
#define VMV_FF_ABLATE 0    // experiments (wrong results): 1 no MFMAs, 2 no fragment reads, 3 GELU -> identity, 4 no chunk barrier / DMA
#endif

VMV_DEV u32x4_t ff_swap16_xz_yw(u32x4_t v) {         // (gemm_rs.hip: padded v_permlane16_swap pair)
    uint32_t x = v.x, y = v.y, z = v.z, w = v.w;
    asm("s_nop 3\n\tv_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3\n\ts_nop 1" : "+v"(x), "+v"(y), "+v"(z), "+v"(w));
    return u32x4_t{x, y, z, w};
}

__global__ __launch_bounds__(512, 1) void ff_fused_kernel(const VmvFfParams p) {
    VMV_KERNEL_ENTER();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fgrp = lane >> 4;
    const int m_wave = blockIdx.x * 128 + wave * 16;
    const int nchunk = 4 * FF_C / FF_HC;             // 40

    // ---- the wave's 16 rows of x, whole K range (rows >= M read as zero through the descriptor)
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (uint32_t)p.M * (uint32_t)p.ldx * 2u, SRD_FLAGS);
    u32x4_t a[FF_KS];
    {
        const uint32_t vo = (uint32_t)((m_wave + frow) * p.ldx + 8 * fgrp) * 2u;
#pragma unroll
        for (int kk = 0; kk < FF_KS; ++kk) a[kk] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, vo + (uint32_t)(kk * 64), 0, 0);
    }

    // ---- loader: bias strips, then the (W1, W2) ring
    float* b1_lds = reinterpret_cast<float*>(smem + 2 * FF_STAGE);
    float* b2_lds = b1_lds + 8 * FF_C;
    {
        const __amdgpu_buffer_rsrc_t b1_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b1), 0, p.b1 ? 8u * FF_C * 4u : 0u, SRD_FLAGS);
        const __amdgpu_buffer_rsrc_t b2_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b2), 0, p.b2 ? FF_C * 4u : 0u, SRD_FLAGS);
        for (int q = wave; q * 256 < 8 * FF_C; q += 8)
            VMV_BLDS16(b1_rsrc, reinterpret_cast<unsigned char*>(b1_lds) + q * 1024, (uint32_t)(q * 256 + 4 * lane) * 4u, 0);
        if (wave < 2) VMV_BLDS16(b2_rsrc, reinterpret_cast<unsigned char*>(b2_lds) + wave * 1024, (uint32_t)(wave * 256 + 4 * lane) * 4u, 0);
    }
    const __amdgpu_buffer_rsrc_t w1_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w1), 0, 8u * FF_C * FF_C * 2u, SRD_FLAGS);
    const __amdgpu_buffer_rsrc_t w2_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w2), 0, (uint32_t)FF_C * 4u * FF_C * 2u, SRD_FLAGS);
    // W2 piece t (16 rows x 64 B): lane -> (row 16 t + (lane >> 2), physical slot lane & 3) fetches k-slot (lane & 3) ^ T[(row >> 2) & 3]
    const int w2_slot = (lane & 3) ^ ((0x78 >> (2 * ((lane >> 4) & 3))) & 3);
    auto issue_chunk = [&](int c, int slot) {
        unsigned char* base = smem + slot * FF_STAGE;
        {   // W1: wave w fills bytes [5120 w, 5120 (w + 1)) of the 40-KB image (64 rows x 40 slots), k-slot s ^ ((r >> 1) & 7) at (r, s)
            const uint32_t so = (uint32_t)(c * 2 * FF_HC * FF_C) * 2u;
            int ln = lane;
            asm volatile("" : "+v"(ln));
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int u = wave * 320 + q * 64 + ln;
                const int r = u / 40, s = u - r * 40;
                VMV_BLDS16(w1_rsrc, base + wave * 5120 + q * 1024, (uint32_t)(r * FF_C + (s ^ ((r >> 1) & 7)) * 8) * 2u, so);
            }
        }
        {   // W2: 20 pieces, wave w takes w, w + 8, w + 16
            const uint32_t so = (uint32_t)(c * FF_HC) * 2u;
            for (int t = wave; t < 20; t += 8)
                VMV_BLDS16(w2_rsrc, base + FF_W1_BYTES + t * 1024, (uint32_t)((16 * t + (lane >> 2)) * (4 * FF_C) + w2_slot * 8) * 2u, so);
        }
    };
    issue_chunk(0, 0);

    // ---- LayerNorm of the resident rows (two-pass, fp32; gemm_rs.hip)
    if (p.ln_eps > 0.f) {
        const float inv_k = 1.0f / (float)FF_C;
        float s1 = 0.f;
#pragma unroll
        for (int kk = 0; kk < FF_KS; ++kk) {
            elem_dot2c(s1, a[kk].x, VMV_ELEM_ONE2); elem_dot2c(s1, a[kk].y, VMV_ELEM_ONE2);
            elem_dot2c(s1, a[kk].z, VMV_ELEM_ONE2); elem_dot2c(s1, a[kk].w, VMV_ELEM_ONE2);
        }
        asm volatile("s_nop 4" : "+v"(s1));          // (DOT result -> VALU read: gemm_rs.hip)
        const float mean = xor16_32_sum(s1) * inv_k;
        float s2 = 0.f;
#pragma unroll
        for (int kk = 0; kk < FF_KS; ++kk) {
            const uint32_t w4[4] = {a[kk].x, a[kk].y, a[kk].z, a[kk].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d0 = elem_lo(w4[e]) - mean, d1 = elem_hi(w4[e]) - mean;
                s2 = fmaf(d0, d0, s2); s2 = fmaf(d1, d1, s2);
            }
        }
        const float rstd = __builtin_amdgcn_rsqf(xor16_32_sum(s2) * inv_k + p.ln_eps);
        const float nm = -mean * rstd;
#pragma unroll
        for (int kk = 0; kk < FF_KS; ++kk) asm volatile("" : "+v"(a[kk].x), "+v"(a[kk].y), "+v"(a[kk].z), "+v"(a[kk].w));
#pragma unroll
        for (int kk = 0; kk < FF_KS; ++kk) {
            a[kk].x = pack_elem2(fmaf(elem_lo(a[kk].x), rstd, nm), fmaf(elem_hi(a[kk].x), rstd, nm));
            a[kk].y = pack_elem2(fmaf(elem_lo(a[kk].y), rstd, nm), fmaf(elem_hi(a[kk].y), rstd, nm));
            a[kk].z = pack_elem2(fmaf(elem_lo(a[kk].z), rstd, nm), fmaf(elem_hi(a[kk].z), rstd, nm));
            a[kk].w = pack_elem2(fmaf(elem_lo(a[kk].w), rstd, nm), fmaf(elem_hi(a[kk].w), rstd, nm));
        }
    }

    f32x4_t out[FF_NT];
#pragma unroll
    for (int j = 0; j < FF_NT; ++j) out[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fragment addressing.  W1 image: rows of 640 B, k-slot (4 kk + fgrp) ^ ((frow >> 1) & 7) (gemm_rs.hip: two lane offsets, one
    // per kk parity, every read base + 64 kk).  W2 image: rows of 64 B, k-slot fgrp ^ T[(frow >> 2) & 3].
    const int fsw = (frow >> 1) & 7;
    int foff[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) foff[r] = frow * 640 + (fgrp ^ (fsw & 3)) * 16 + ((r ^ (fsw >> 2)) - r) * 64;
    const int f2off = frow * 64 + (fgrp ^ ((0x78 >> (2 * ((frow >> 2) & 3))) & 3)) * 16;

    for (int c = 0; c < nchunk; ++c) {
        const int slot = c & 1;
        // chunk c landed (mine: nothing but DMAs are in flight), then for every wave; every wave is also done with chunk c - 1,
        // whose slot takes chunk c + 1 — which then has the whole of chunk c's 60 MFMAs per wave to arrive
#if VMV_FF_ABLATE == 4
        if (c == 0) { wait_vmcnt<0>(); __builtin_amdgcn_s_barrier(); }
#else
        wait_vmcnt<0>();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (c + 1 < nchunk) issue_chunk(c + 1, slot ^ 1);
#endif
        const unsigned char* w1b = smem + slot * FF_STAGE;
        const unsigned char* w2b = w1b + FF_W1_BYTES + f2off;

        // ---- FF1: tiles (x0, g0, x1, g1) = rows [0, 16), [16, 32), [32, 48), [48, 64) of the chunk's W1 image
        f32x4_t h1[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) h1[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        constexpr int PF1 = VMV_FF_PF1, NB1 = PF1 + 1;
        u32x4_t wf[NB1][4];
        auto rd1 = [&](const int kk, u32x4_t (&w)[4]) {
            const unsigned char* tp = w1b + foff[kk & 1] + 64 * kk;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#if VMV_FF_ABLATE == 2
                w[t] = u32x4_t{(uint32_t)(uintptr_t)tp, 1u, 2u, (uint32_t)t};
#else
                w[t] = *reinterpret_cast<const u32x4_t*>(tp + t * (16 * 640));
#endif
            }
        };
#pragma unroll
        for (int kk = 0; kk < PF1; ++kk) rd1(kk, wf[kk % NB1]);
#pragma unroll
        for (int kk = 0; kk < FF_KS; ++kk) {
            if (kk + PF1 < FF_KS) rd1(kk + PF1, wf[(kk + PF1) % NB1]);
            __builtin_amdgcn_sched_barrier(0);
#if VMV_FF_ABLATE == 1
            h1[kk & 3].x += __uint_as_float(wf[kk % NB1][0].x ^ wf[kk % NB1][1].y ^ wf[kk % NB1][2].z ^ wf[kk % NB1][3].w ^ a[kk].x);
#else
#pragma unroll
            for (int t = 0; t < 4; ++t)
                h1[t] = VMV_MFMA16(__builtin_bit_cast(elem8_t, wf[kk % NB1][t]), __builtin_bit_cast(elem8_t, a[kk]), h1[t], 0, 0, 0);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- W2 fragments of the first output tiles go out under the GELU
        constexpr int PF2 = VMV_FF_PF2, NB2 = PF2 + 1;
        u32x4_t w2f[NB2];
        auto rd2 = [&](const int j) -> u32x4_t {
#if VMV_FF_ABLATE == 2
            return u32x4_t{(uint32_t)(uintptr_t)w2b, 1u, 2u, (uint32_t)j};
#else
            return *reinterpret_cast<const u32x4_t*>(w2b + j * (16 * 64));
#endif
        };
#pragma unroll
        for (int j = 0; j < PF2; ++j) w2f[j % NB2] = rd2(j);
        // ---- GEGLU of the two hidden tiles -> the B fragment of this chunk's k-step
        u32x4_t hf;
        {
            const float* bb = b1_lds + c * 64 + 4 * fgrp;
            f32x4_t x0 = h1[0] + *reinterpret_cast<const f32x4_t*>(bb);
            const f32x4_t g0 = h1[1] + *reinterpret_cast<const f32x4_t*>(bb + 16);
            f32x4_t x1 = h1[2] + *reinterpret_cast<const f32x4_t*>(bb + 32);
            const f32x4_t g1 = h1[3] + *reinterpret_cast<const f32x4_t*>(bb + 48);
#if VMV_FF_ABLATE == 3
            x0 *= g0; x1 *= g1;
#else
            x0.x *= gelu_erf_f(g0.x); x0.y *= gelu_erf_f(g0.y); x0.z *= gelu_erf_f(g0.z); x0.w *= gelu_erf_f(g0.w);
            x1.x *= gelu_erf_f(g1.x); x1.y *= gelu_erf_f(g1.y); x1.z *= gelu_erf_f(g1.z); x1.w *= gelu_erf_f(g1.w);
#endif
            hf.x = pack_elem2(x0.x, x0.y); hf.y = pack_elem2(x0.z, x0.w);
            hf.z = pack_elem2(x1.x, x1.y); hf.w = pack_elem2(x1.z, x1.w);
        }
        // ---- FF2: out[16 rows][320] += h[16][32] . W2[320][32]^T, one MFMA per output tile; fragments three tiles ahead
#pragma unroll
        for (int j = 0; j < FF_NT; ++j) {
            if (j + PF2 < FF_NT) w2f[(j + PF2) % NB2] = rd2(j + PF2);
            __builtin_amdgcn_sched_barrier(0);
#if VMV_FF_ABLATE == 1
            out[j].x += __uint_as_float(w2f[j % NB2].x ^ hf.x);
#else
            out[j] = VMV_MFMA16(__builtin_bit_cast(elem8_t, w2f[j % NB2]), __builtin_bit_cast(elem8_t, hf), out[j], 0, 0, 0);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- epilogue: + b2 + residual, 16-byte stores through the lane swap of gemm_rs.hip (two output tiles per store)
    const int lanecol = (fgrp & 1) * 16 + (fgrp >> 1) * 8;
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, SRD_RECORDS, SRD_FLAGS);
    const __amdgpu_buffer_rsrc_t res_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.residual), 0, SRD_RECORDS, SRD_FLAGS);
    const bool ok = m_wave + frow < p.M;
    const uint32_t ovo = ok ? (uint32_t)((m_wave + frow) * p.ldo + lanecol) * 2u : OOB;
    const uint32_t rvo = (ok && p.residual) ? (uint32_t)((m_wave + frow) * p.ldr + lanecol) * 2u : OOB;
#pragma unroll
    for (int jp = 0; jp < FF_NT / 2; ++jp) {
        const u32x4_t rr = ff_swap16_xz_yw(__builtin_amdgcn_raw_buffer_load_b128(res_rsrc, rvo, (uint32_t)(32 * jp) * 2u, 0));
        f32x4_t v0 = out[2 * jp] + *reinterpret_cast<const f32x4_t*>(b2_lds + 32 * jp + 4 * fgrp);
        f32x4_t v1 = out[2 * jp + 1] + *reinterpret_cast<const f32x4_t*>(b2_lds + 32 * jp + 16 + 4 * fgrp);
        v0.x += elem_lo(rr.x); v0.y += elem_hi(rr.x); v0.z += elem_lo(rr.y); v0.w += elem_hi(rr.y);
        v1.x += elem_lo(rr.z); v1.y += elem_hi(rr.z); v1.z += elem_lo(rr.w); v1.w += elem_hi(rr.w);
        u32x4_t o;
        o.x = pack_elem2(v0.x, v0.y); o.y = pack_elem2(v0.z, v0.w);
        o.z = pack_elem2(v1.x, v1.y); o.w = pack_elem2(v1.z, v1.w);
        o = ff_swap16_xz_yw(o);
        __builtin_amdgcn_raw_buffer_store_b128(o, out_rsrc, ovo, (uint32_t)(32 * jp) * 2u, 0);
        asm volatile("s_nop 7" ::"v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w) : "memory");       // (store-data discipline: gemm_rs.hip)
    }
}

}  // namespace

extern "C" int vmv_ff_fused_ok(const VmvFfParams* pp) {
    if (!pp) return 0;
    const VmvFfParams& p = *pp;
    if (p.C != FF_C || p.M <= 0 || !p.x || !p.w1 || !p.w2 || !p.out) return 0;
    if ((p.ldx & 7) || (p.ldo & 7) || (p.residual && (p.ldr & 7))) return 0;
    if (!vmv_aligned16(p.x) || !vmv_aligned16(p.w1) || !vmv_aligned16(p.w2) || !vmv_aligned16(p.out) || (p.residual && !vmv_aligned16(p.residual))) return 0;
    if ((p.b1 && !vmv_aligned16(p.b1)) || (p.b2 && !vmv_aligned16(p.b2))) return 0;
    if ((long)(p.M + 128) * p.ldx * 2 >= (1L << 31) - 65536 || (long)(p.M + 128) * p.ldo * 2 >= (1L << 31) - 65536) return 0;
    if (p.residual && (long)(p.M + 128) * p.ldr * 2 >= (1L << 31) - 65536) return 0;
    return 1;
}

extern "C" int vmv_ff_fused(const VmvFfParams* pp, void* stream) {
    if (!pp) return VMV_ENULL;
    const VmvFfParams& p = *pp;
    if (!p.x || !p.w1 || !p.w2 || !p.out) return VMV_ENULL;
    if (p.C != FF_C || p.M <= 0) return VMV_EINVAL;
    if (!vmv_ff_fused_ok(pp)) return VMV_EALIGN;
    static std::atomic<unsigned long long> attr{0};
    if (const int rc_attr = vmv_lds_attr_once(attr, reinterpret_cast<const void*>(&ff_fused_kernel), FF_LDS)) return rc_attr;
    hipLaunchKernelGGL(ff_fused_kernel, dim3((p.M + 127) / 128), dim3(512), FF_LDS, reinterpret_cast<hipStream_t>(stream), p);
    return vmv_launch_status();
}

// gemm_astat.hip — A-STATIONARY persistent variant of the LDS-DMA MFMA GEMM for the short-K, wide-N linears of the
// transformer blocks at the largest level (K = C = 320: LayerNorm-folded qkv / q and the GEGLU up-projection, N = 3C .. 8C).
//
// Why (measured with VMV_GEMM_ABLATE on these shapes): with K = 320 a 256 x 160 output tile has only 5
// chunks of main loop, and the tile-per-item kernels pay per tile (a) a fresh A tile through the ring although the next
// N tile needs the very same rows, (b) an epilogue that idles the MFMA pipes while 80 KB are packed and stored, (c) a ring
// restart: removing the MFMAs altogether only made them 17 % faster.  Here a block owns a PANEL of 128 rows for all of N:
//   * the A panel [128 x K] is DMA'd into LDS once per panel (K / 64 chunks of 16 KB) and stays there while the block
//     walks the N tiles; only weights stream through the 3-stage ring (W is L2-resident: every CU reads the same 160 rows
//     at about the same time), so the vector-memory path carries 20 KB per chunk instead of 52;
//   * the K loop never restarts: (panel, N tile, chunk) is one flat sequence of steps, the ring runs two steps ahead across
//     N-tile and panel boundaries, and the next panel's A chunks replace the current ones one step after their last use;
//   * the epilogue is DEFERRED: two accumulator sets (40 registers each at a 32 x 80 wave tile); while the MFMAs of N tile
//     j + 1 run on one set, the other is drained — LN-fold / bias / GEGLU math and 8-byte stores spread over the 5 chunks —
//     so the matrix pipes never wait for an epilogue; its inputs (bias, column sums, the panel's row statistics) arrive by
//     LDS-DMA as well, so no VGPR-destination load ever forces a vmcnt drain of the ring.
// Block = 8 waves (4 along M x 2 along N), wave tile 32 x 16 WN, same LDS image / swizzle / transposed-product fragment
// layout as gemm_glds.hip.  Contract: one LINEAR segment, K = 64 KC (KC <= 5), 16-bit output, no residual / rowvec / act
// (the dispatcher sends everything else to the other variants).
#include "gemm_glds_common.h"
#include <cstdlib>
#include <type_traits>

using namespace vmvg;

namespace {

template <int WN, int KC>
struct AsCfg {
    static constexpr int NW = 8, NT = 512, BM = 128, WM = 2;
    static constexpr int BN = 32 * WN;
    static constexpr int A_CHUNK = BM * 128;
    static constexpr int A_BYTES = KC * A_CHUNK;
    static constexpr int W_CHUNK = BN * 128;
    static constexpr int STAGES = 3;
    static constexpr int W_BYTES = STAGES * W_CHUNK;
    // per wave and N tile: 16 WN bias values | 16 WN column sums, each region padded to whole 64-lane DMA instructions
    // (out-of-range lanes of an LDS-DMA write zeros: the padding absorbs them)
    static constexpr int NSTRIP = (16 * WN + 63) / 64;
    static constexpr int REGION = NSTRIP * 256;
    static constexpr int STRIP = 2 * REGION;
    static constexpr int STRIPS = 2 * NW * STRIP;            // double-buffered (N tile parity)
    static constexpr int RSTAT = 2 * NW * 256;               // per wave: (mean, rstd) of its 32 rows, double-buffered (panel parity)
    static constexpr int LDS_TOTAL = A_BYTES + W_BYTES + STRIPS + RSTAT;
    static constexpr int NAI = BM / (8 * NW);                // A pieces (8 rows x 128 B) per wave per chunk = 2
    static constexpr int NWI = (BN / 8 + NW - 1) / NW;       // W pieces per wave per chunk
    static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");
};

VMV_DEV void wait_vmcnt_n(int n) {
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
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;      // (stronger than needed: still correct)
    }
}

template <int WN, int KC, bool GEGLU, int ablate = 0>
__global__ __launch_bounds__(512) void gemm_astat_kernel(const VmvGemmParams p, const int npanels, const int ntn) {
    VMV_KERNEL_ENTER();
    using Cfg = AsCfg<WN, KC>;
    constexpr int NW = Cfg::NW, WM = Cfg::WM, BN = Cfg::BN;
    static_assert(!GEGLU || (WN % 2) == 0, "GEGLU pairs x / gate column tiles");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const ring = smem + Cfg::A_BYTES;
    unsigned char* const strips = ring + Cfg::W_BYTES;
    unsigned char* const rstats = strips + Cfg::STRIPS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int G = gridDim.x, bid = blockIdx.x;
    const int my_panels = bid < npanels ? (npanels - 1 - bid) / G + 1 : 0;
    if (my_panels == 0) return;
    const int NS = my_panels * ntn;                 // N-tile steps of this block
    const int T = NS * KC;                          // chunk steps

    const int lrow = lane >> 3;
    const int lsw = (lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);     // logical 16-B slot this lane fetches
    const int frow = lane & 15, fgrp = lane >> 4, fswz = (frow >> 1) & 7;
    const int K = p.ktot;
    const VmvGemmSeg& sg = p.seg[0];
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(sg.src), 0, SRD_RECORDS, SRD_FLAGS);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, SRD_RECORDS, SRD_FLAGS);
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, SRD_RECORDS, SRD_FLAGS);
    const __amdgpu_buffer_rsrc_t bias_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias ? p.bias : p.colsum), 0, (uint32_t)p.N * 4u, SRD_FLAGS);
    const __amdgpu_buffer_rsrc_t csum_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.colsum ? p.colsum : p.bias), 0, (uint32_t)p.N * 4u, SRD_FLAGS);
    const __amdgpu_buffer_rsrc_t rs_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.rowstat), 0, (uint32_t)p.M * 8u, SRD_FLAGS);
    const bool has_bias = p.bias != nullptr, has_ln = p.rowstat != nullptr;
    const int N_out = GEGLU ? p.N / 2 : p.N;

    // ---- per-panel loader state
    int wgrp[Cfg::NWI];
#pragma unroll
    for (int j = 0; j < Cfg::NWI; ++j) {
        int g = j * NW + wave;
        if (g >= BN / 8) g -= NW;                   // duplicate an earlier row group: uniform load count per wave
        wgrp[j] = g;
    }
    auto a_offsets = [&](int m0, uint32_t (&avo)[Cfg::NAI]) {
#pragma unroll
        for (int i = 0; i < Cfg::NAI; ++i) {
            const int m = m0 + (i * NW + wave) * 8 + lrow;
            avo[i] = (m < p.M) ? (uint32_t)(m * sg.ld + lsw * 8) * 2u : OOB;
        }
    };
    auto issue_a = [&](const uint32_t (&avo)[Cfg::NAI], int kc) {          // A chunk kc of a panel -> its resident slot
        unsigned char* base = smem + kc * Cfg::A_CHUNK + wave * 1024;
#pragma unroll
        for (int i = 0; i < Cfg::NAI; ++i) VMV_BLDS16(a_rsrc, base + i * (NW * 1024), avo[i], (uint32_t)kc * 128u);
    };
    auto issue_w = [&](int n0, int kc, int slot) {                          // W chunk (n0, kc) -> ring slot
        unsigned char* base = ring + slot * Cfg::W_CHUNK;
#pragma unroll
        for (int j = 0; j < Cfg::NWI; ++j) {
            const int n = n0 + wgrp[j] * 8 + lrow;
            const uint32_t vo = (n < p.N) ? (uint32_t)(n * K + lsw * 8) * 2u : OOB;
            VMV_BLDS16(w_rsrc, base + wgrp[j] * 1024, vo, (uint32_t)kc * 128u);
        }
    };
    constexpr int NSTRIP = Cfg::NSTRIP;                                     // 4-byte DMA instructions per strip region
    auto issue_strips = [&](int n0, int par) -> int {                       // bias | colsum of this wave's 16 WN columns
        unsigned char* sb = strips + (par * NW + wave) * Cfg::STRIP;
        int cnt = 0;
#pragma unroll
        for (int r = 0; r < NSTRIP; ++r) {
            const int c = lane + 64 * r;
            const int n = n0 + wave_n * 16 * WN + c;
            const uint32_t vo = (c < 16 * WN && n < p.N) ? (uint32_t)n * 4u : OOB;
            if (has_bias) { blds4(bias_rsrc, sb + r * 256, vo, 0); ++cnt; }
            if (has_ln) { blds4(csum_rsrc, sb + Cfg::REGION + r * 256, vo, 0); ++cnt; }
        }
        return cnt;
    };
    auto issue_rstat = [&](int m0, int par) -> int {                        // (mean, rstd) of this wave's 32 rows: 64 floats
        if (!has_ln) return 0;
        const int m = m0 + wave_m * 32 + (lane >> 1);
        const uint32_t vo = (m < p.M) ? (uint32_t)(m * 2 + (lane & 1)) * 4u : OOB;
        blds4(rs_rsrc, rstats + (par * NW + wave) * 256, vo, 0);
        return 1;
    };

    // ---- accumulators (two sets: one computes, the other drains) and the deferred-epilogue state
    f32x4_t acc0[WN][WM], acc1[WN][WM];
    struct EpiState { uint32_t rowoff[WM]; int n0; int spar; int rpar; };
    EpiState E;
    E.n0 = 0; E.spar = 0; E.rpar = 0;
#pragma unroll
    for (int i = 0; i < WM; ++i) E.rowoff[i] = OOB;
    constexpr int UNITS = GEGLU ? WM * (WN / 2) : WM * WN;
    // unit u of a drained N tile: row group i = u % WM, column tile j (GEGLU: x tile 2 * (u / WM), gate tile + 1)
    auto epi_unit = [&](const f32x4_t (&accP)[WN][WM], const EpiState& e, const int u) {
        const int i = u % WM;
        const int j = GEGLU ? 2 * (u / WM) : (u / WM);
        const float* sb = reinterpret_cast<const float*>(strips + (e.spar * NW + wave) * Cfg::STRIP);
        f32x4_t v = accP[j][i];
        f32x4_t g = GEGLU ? accP[GEGLU ? j + 1 : j][i] : v;
        if (has_ln) {
            const float2 ms = *reinterpret_cast<const float2*>(rstats + (e.rpar * NW + wave) * 256 + (16 * i + frow) * 8);
            const f32x4_t c0 = *reinterpret_cast<const f32x4_t*>(sb + Cfg::REGION / 4 + 16 * j + 4 * fgrp);
            v = (v - c0 * ms.x) * ms.y;
            if constexpr (GEGLU) {
                const f32x4_t c1 = *reinterpret_cast<const f32x4_t*>(sb + Cfg::REGION / 4 + 16 * (j + 1) + 4 * fgrp);
                g = (g - c1 * ms.x) * ms.y;
            }
        }
        if (has_bias) {
            v += *reinterpret_cast<const f32x4_t*>(sb + 16 * j + 4 * fgrp);
            if constexpr (GEGLU) g += *reinterpret_cast<const f32x4_t*>(sb + 16 * (j + 1) + 4 * fgrp);
        }
        int no;
        if constexpr (GEGLU) {
            v.x *= gelu_erf_f(g.x); v.y *= gelu_erf_f(g.y); v.z *= gelu_erf_f(g.z); v.w *= gelu_erf_f(g.w);
            no = e.n0 / 2 + wave_n * 8 * WN + 8 * j + 4 * fgrp;            // x tile j -> output tile j / 2
        } else {
            no = e.n0 + wave_n * 16 * WN + 16 * j + 4 * fgrp;
        }
        u32x2_t o;
        o.x = pack_elem2(v.x, v.y); o.y = pack_elem2(v.z, v.w);
        const uint32_t vo = (e.rowoff[i] != OOB && no < N_out) ? e.rowoff[i] + (uint32_t)no * 2u : OOB;
        if constexpr (ablate != 7) __builtin_amdgcn_raw_buffer_store_b64(o, out_rsrc, vo, 0, 0);
        else if (o.x == 0x12345u && vo == 77u) __builtin_amdgcn_raw_buffer_store_b64(o, out_rsrc, vo, 0, 0);   // (keeps the value alive)
    };

    // ---- prologue: panel 0's A chunks, statistics and strips, W chunks of steps 0 and 1
    int panel = bid;                                 // panel of the COMPUTE position
    uint32_t avo[Cfg::NAI], avo_next[Cfg::NAI];
    a_offsets(panel * Cfg::BM, avo);
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) issue_a(avo, kc);
    issue_rstat(panel * Cfg::BM, 0);
    issue_strips(0, 0);
    // loader position = compute position + 2 steps
    int L_n0 = 0, L_kc = 0, L_nt = 0, L_left = T;   // next W chunk to issue
    auto advance_loader = [&]() {
        --L_left;
        if (++L_kc == KC) { L_kc = 0; L_n0 += BN; if (++L_nt == ntn) { L_nt = 0; L_n0 = 0; } }
    };
    issue_w(L_n0, L_kc, 0); advance_loader();
    int issued_w2 = 0;
    if (L_left > 0) { issue_w(L_n0, L_kc, 1); advance_loader(); issued_w2 = Cfg::NWI; }
    wait_vmcnt_n(issued_w2);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int slot = 0;                                    // ring slot of the chunk to consume
    int after_w_prev = 0;                            // memory ops issued after the W group of the previous step
    int g_step = 0;
    uint32_t rowoff_cur[WM];
    auto set_rows = [&](int m0) {
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int m = m0 + wave_m * 32 + 16 * i + frow;
            rowoff_cur[i] = (m < p.M) ? (uint32_t)(m * p.ldo) * 2u : OOB;
        }
    };
    set_rows(panel * Cfg::BM);
    int pidx = 0;                                    // local panel index (parity selects the statistics buffer)
    int nt = 0;

    // One N-tile step: KC chunk steps on accC while accP (the previous N tile, if `drain`) is written out.
    auto nstep = [&](f32x4_t (&accC)[WN][WM], const f32x4_t (&accP)[WN][WM], const bool drain, const int ns) {
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int i = 0; i < WM; ++i) accC[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const bool last_nt = nt == ntn - 1;
        const bool have_next_panel = pidx + 1 < my_panels;
        if (last_nt && have_next_panel) a_offsets((panel + G) * Cfg::BM, avo_next);
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            // ---- (barrier passed: W chunk of this step and the A chunks it needs are visible)
            int cur = 0;
            // A refresh: chunk kc - 1 of the NEXT panel once its slot's last reader (previous step) is past the barrier
            if (kc >= 1 && last_nt && have_next_panel) { issue_a(avo_next, kc - 1); cur += Cfg::NAI; }
            if (kc == 0 && nt == 0 && pidx > 0) { issue_a(avo, KC - 1); cur += Cfg::NAI; }     // last chunk of THIS panel
            if (kc == 0) {
                if (nt == 0 && pidx > 0) cur += issue_rstat(panel * Cfg::BM, pidx & 1);
                if (ns > 0) cur += issue_strips(nt * BN, ns & 1);          // (step 0's went out in the prologue)
            }
            if (L_left > 0) {
                int s2 = slot + 2; if (s2 >= Cfg::STAGES) s2 -= Cfg::STAGES;
                issue_w(L_n0, L_kc, s2); advance_loader(); cur += Cfg::NWI;
            }
            asm volatile("" ::: "memory");
            // ---- MFMAs of this chunk (+ the drained tile's epilogue units of this chunk)
            const u32x4_t* a = reinterpret_cast<const u32x4_t*>(smem + kc * Cfg::A_CHUNK) + (wave_m * 32 + frow) * 8;
            const u32x4_t* w = reinterpret_cast<const u32x4_t*>(ring + slot * Cfg::W_CHUNK) + (wave_n * 16 * WN + frow) * 8;
            elem8_t af[2][WM], wf[2][WN];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int sl = (kk * 4 + fgrp) ^ fswz;
#pragma unroll
                for (int i = 0; i < WM; ++i) af[kk][i] = __builtin_bit_cast(elem8_t, a[i * 16 * 8 + sl]);
#pragma unroll
                for (int j = 0; j < WN; ++j) wf[kk][j] = __builtin_bit_cast(elem8_t, w[j * 16 * 8 + sl]);
            }
            int stores = 0;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int j = 0; j < WN; ++j)
#pragma unroll
                    for (int i = 0; i < WM; ++i) accC[j][i] = VMV_MFMA16(wf[kk][j], af[kk][i], accC[j][i], 0, 0, 0);
                if (drain) {
                    // units [ (2 kc + kk) * UNITS / (2 KC), (2 kc + kk + 1) * UNITS / (2 KC) ) of the previous N tile
                    constexpr int H = 2 * KC;
                    const int h = 2 * kc + kk;
#pragma unroll
                    for (int u = 0; u < UNITS; ++u)
                        if (u >= (h * UNITS) / H && u < ((h + 1) * UNITS) / H) { epi_unit(accP, E, u); ++stores; }
                }
            }
            asm volatile("" ::: "memory");
            // ---- next step's chunk landed (mine); everything issued after it may stay in flight
            ++g_step;
            if (g_step < T) {
                wait_vmcnt_n(after_w_prev + cur + stores);
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            after_w_prev = stores;
            slot = slot + 1 == Cfg::STAGES ? 0 : slot + 1;
        }
        // latch the epilogue state of the tile just computed
#pragma unroll
        for (int i = 0; i < WM; ++i) E.rowoff[i] = rowoff_cur[i];
        E.n0 = nt * BN; E.spar = ns & 1; E.rpar = pidx & 1;
        if (++nt == ntn) {
            nt = 0;
            if (++pidx < my_panels) {
                panel += G;
#pragma unroll
                for (int i = 0; i < Cfg::NAI; ++i) avo[i] = avo_next[i];
                set_rows(panel * Cfg::BM);
            }
        }
    };

    for (int ns = 0; ns < NS; ns += 2) {
        nstep(acc0, acc1, ns > 0, ns);
        if (ns + 1 < NS) nstep(acc1, acc0, true, ns + 1);
    }
    // ---- drain the last N tile
    asm volatile("" ::: "memory");
    if (NS & 1) {
#pragma unroll
        for (int u = 0; u < UNITS; ++u) epi_unit(acc0, E, u);
    } else {
#pragma unroll
        for (int u = 0; u < UNITS; ++u) epi_unit(acc1, E, u);
    }
}

template <int WN, int KC, bool GEGLU>
int launch_astat(const VmvGemmParams& p, hipStream_t st) {
    using Cfg = AsCfg<WN, KC>;
    const int npanels = (p.M + Cfg::BM - 1) / Cfg::BM;
    const int ntn = (p.N + Cfg::BN - 1) / Cfg::BN;
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0, n = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return (int)e;
        ncu = n < 1 ? 1 : n;
    }
    static int ablate = -1;
    if (ablate < 0) { const char* e = getenv("VMV_GEMM_ABLATE"); ablate = e ? atoi(e) : 0; }
    const int G = npanels < ncu ? npanels : ncu;
    auto go = [&](auto tag) -> int {
        constexpr int AB = decltype(tag)::value;
        static std::atomic<unsigned long long> attr_set{0};
        if (const int rc_attr = vmv_lds_attr_once(attr_set, reinterpret_cast<const void*>(&gemm_astat_kernel<WN, KC, GEGLU, AB>), Cfg::LDS_TOTAL)) return rc_attr;
        hipLaunchKernelGGL((gemm_astat_kernel<WN, KC, GEGLU, AB>), dim3(G), dim3(Cfg::NT), Cfg::LDS_TOTAL, st, p, npanels, ntn);
        return VMV_OK;
    };
    int rc;
    if (ablate == 7) rc = go(std::integral_constant<int, 7>{});
    else if (ablate == 1) rc = go(std::integral_constant<int, 1>{});
    else rc = go(std::integral_constant<int, 0>{});
    if (rc != VMV_OK) return rc;
    return vmv_launch_status();
}

}  // namespace

// Eligibility (also used by pick_tile): one LINEAR segment, K in {256, 320}, 16-bit output, no residual / rowvec / act /
// split-K, at least 2 N tiles, operands addressable with 32-bit buffer offsets.
bool vmv_gemm_astat_eligible(const VmvGemmParams& p) {
    if (p.nseg != 1 || p.seg[0].mode != VMV_SEG_LINEAR) return false;
    if (p.ktot != 320 && p.ktot != 256) return false;
    if (p.out_fp32 || p.residual || p.rowvec || p.act != VMV_ACT_NONE || p.ksplit > 1) return false;
    if (p.rowstat && !p.colsum) return false;
    const bool geglu = p.epilogue == VMV_EPI_GEGLU;
    if (geglu && (p.N % 32)) return false;
    if ((p.ldo & 3) || (((uintptr_t)p.out) & 7)) return false;
    if ((long)(p.M + 128) * p.seg[0].ld * 2 >= (1L << 31) - 65536) return false;
    if ((long)p.N * p.ktot * 2 >= (1L << 31) - 65536) return false;
    if ((long)(p.M + 128) * p.ldo * 2 >= (1L << 31) - 65536) return false;
    if (p.bias && (((uintptr_t)p.bias) & 3)) return false;
    return true;
}

int vmv_gemm_astat_launch(const VmvGemmParams& p, int tile, hipStream_t st) {
    if (!vmv_gemm_astat_eligible(p)) return VMV_GLDS_UNSUPPORTED;
    const bool geglu = p.epilogue == VMV_EPI_GEGLU;
    if (tile == VMV_TILE_A128x160) {
        if (geglu) return VMV_EINVAL;
        return p.ktot == 320 ? launch_astat<5, 5, false>(p, st) : launch_astat<5, 4, false>(p, st);
    }
    if (tile == VMV_TILE_A128x128) {
        if (geglu) return p.ktot == 320 ? launch_astat<4, 5, true>(p, st) : launch_astat<4, 4, true>(p, st);
        return p.ktot == 320 ? launch_astat<4, 5, false>(p, st) : launch_astat<4, 4, false>(p, st);
    }
    return VMV_EINVAL;
}

// attention.hip — softmax(Q K^T * scale) V for head_dim 64 on gfx950, bf16 in/out, fp32 softmax + accumulate.
// One flash-style kernel serves the UNet's three attention shapes through index maps (include/vmv.h):
// spatial self (N = H*W), spatial cross (Nk = 77 text tokens) and per-pixel temporal (N = 24 frames).
//
// Formulation (everything transposed so that softmax statistics are lane-local):
//   S^T = K Q^T  : MFMA A = K fragment (rows = keys), B = Q fragment (cols = queries)
//                  -> lane (u = lane&15, g = lane>>4) holds S[q = u][4 keys] per 16-key tile
//   O^T = V^T P^T: MFMA A = V^T fragment (rows = d),  B = P^T fragment (cols = queries)
//                  -> lane holds O[q = u][d = 16*dt + 4*g + r]: the same q as its softmax state, so the
//                     running max / sum / rescale never cross lanes, and the output store is 8 bytes per lane.
// The K rows of each 16-key MFMA tile are loaded in a permuted order (tile t, row u -> key 32*(t>>1) + 8*(u>>2)
// + 4*(t&1) + (u&3)) so that the exponentiated S^T registers of tiles (2kk, 2kk+1) ARE the P^T B-operand for
// keys 32kk + 8g + 0..7 — no shuffles and no LDS round trip between the two GEMMs.
// K tiles are staged in LDS row-major [key][64] and V tiles transposed [d][key] (both XOR-swizzled so that the
// 16-byte fragment reads and the staging writes are bank-conflict free).
//   WPP = 4: the 4 waves of a block share one (problem, 128-query tile) and one K/V stage;
//   WPP = 1: every wave owns a whole short problem (Nq <= 32, e.g. the 24-frame temporal attention) with a
//            private 16-KB stage.
#include "common.h"
#include "gemm_glds_common.h"       // LDS-DMA helper, buffer descriptor constants
#include <cstdlib>

namespace {

VMV_DEV long seq_base(const VmvSeqMap& m, int o) {
    return (long)(o / m.inner) * m.s_outer + (long)(o % m.inner) * m.s_inner;
}
VMV_DEV int k_swz(int key) { return (((key >> 3) & 3) << 1) | ((key >> 1) & 1); }

constexpr float NEG_BIG = -1.0e30f;
#ifndef VMV_ATTN_STAGES
#define VMV_ATTN_STAGES 2   // K / V ring depth of the 4-waves-per-problem kernels (head_dim 32 / 64).  3 (DMA of tile kt + 2 issued in tile kt's
                            // softmax phase, a tile to land) measured SLOWER on one box: L0 self 592 -> 617-627 us, L1 80 -> 96, cross 47 -> 56
#endif
#ifndef VMV_ATTN_ABLATE
#define VMV_ATTN_ABLATE 0   // experiments (tools/experiments/run_attn_ablate.sh): 1 no exp, 2 no softmax VALU, 3 no PV MFMAs, 4 no K/V restaging, 5 = 4 + no barrier
#endif

// QT = 16-query tiles per wave (QT = 4: 64 queries per wave, 256 per block — every K / V fragment read from LDS and
// every staged K / V tile then serves twice the MFMAs; the LDS port, shared by all the blocks of a CU, is what bounds the
// 128-query version at long sequences)
// D = head dimension: 64 (the video UNet) or 32 (the LGM U-Net's 512-channel / 16-head MVAttention; WPP = 4 only): the
// same kernel with D / 32 k-steps in S^T = K Q^T, D / 16 output tiles in O^T = V^T P^T and D / 8 16-byte slots per staged K row.
// CAUSAL: keys j > query i masked out (own instantiation: the test costs <4, 4, 64> two spilled registers)
template <int WPP, int QT, int D = 64, bool CAUSAL = false>
__global__ __launch_bounds__(256, 2) void attn_kernel(const VmvAttnParams p, const int nproblems) {
    VMV_KERNEL_ENTER();
    static_assert(D == 64 || ((D == 32 || D == 128) && WPP == 4), "head_dim 64, or 32 / 128 on the 4-waves-per-problem variant");
    constexpr int KK = D / 32, DT = D / 16, SLOTS = D / 8, SLOG = (D == 128) ? 4 : (D == 64) ? 3 : 2;
    constexpr int KBYTES = (D == 128) ? 16384 : 8192, STAGE = 2 * KBYTES;       // staged K tile (then the V tile) / one stage
    // WPP = 4 ring depth (VMV_ATTN_STAGES, default 2).  Three stages (48 KB): the DMA of tile kt + 2 goes out after tile kt's softmax
    // and has a whole tile to land; with two stages it goes out at the top of the tile, among the S^T MFMAs — which measured faster.
    constexpr int NST = (WPP == 4 && D != 128) ? VMV_ATTN_STAGES : 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = (WPP == 4) ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
    const int u = lane & 15, g = lane >> 4;

    // ---- which problem / query rows does this wave own?
    int o, h, q0;
    bool wvalid = true;
    if constexpr (WPP == 4) {
#ifndef VMV_ATTN_XCD_MAP
#define VMV_ATTN_XCD_MAP 1
#endif
#if VMV_ATTN_XCD_MAP
        // XCD-aware block map: workgroups are handed out round-robin over the 8 XCDs in linear block order, so with the natural map
        // the query tiles that share one (problem, head)'s K / V land on eight different L2s and each re-fetches them from the fabric
        // (FETCH_SIZE: 1.2 GB per L0 self-attention launch for 236 MB of operands).  Bijection: the blocks of one XCD, in issue
        // order, walk consecutive logical ids = the query tiles of one (problem, head), then the next pair.
        const int nqt = gridDim.x, nh = gridDim.y;
        const int lin = blockIdx.x + nqt * (blockIdx.y + nh * blockIdx.z), nblk = nqt * nh * gridDim.z;
        const int qq = nblk >> 3, rr = nblk & 7, xcd = lin & 7, idx = lin >> 3;
        const int logical = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
        const int pair = logical / nqt, qtile = logical - pair * nqt;
        o = pair / nh; h = pair - o * nh; q0 = (qtile * 4 + wave) * (16 * QT);
#else
        o = blockIdx.z; h = blockIdx.y; q0 = (blockIdx.x * 4 + wave) * (16 * QT);
#endif
    } else {
        const int pidx = blockIdx.x * 4 + wave;
        wvalid = pidx < nproblems;
        const int pi = wvalid ? pidx : 0;
        h = pi % p.heads; o = pi / p.heads; q0 = 0;
    }
    const uint16_t* qp = reinterpret_cast<const uint16_t*>(p.q) + seq_base(p.qm, o) + h * D;
    const uint16_t* kp = reinterpret_cast<const uint16_t*>(p.k) + seq_base(p.km, o / p.kv_div) + h * D;
    const uint16_t* vp = reinterpret_cast<const uint16_t*>(p.v) + seq_base(p.vm, o / p.kv_div) + h * D;
    uint16_t* op = reinterpret_cast<uint16_t*>(p.o) + seq_base(p.om, o) + h * D;
    // (o, h) are wave-uniform (block indices, or blockIdx * 4 + wave), so the four problem base pointers are too — but not
    // provably so for the compiler (threadIdx >> 6), which kept them in 8 VGPRs and spilled four of them across the main loop of
    // <4, 4, 64>.  Through readfirstlane they live in SGPRs and every access becomes scalar base + 32-bit lane offset.
    auto uniform_ptr = [](auto* ptr) {
        const uint64_t a = reinterpret_cast<uint64_t>(ptr);
        // (the builtin returns a signed int: widen through uint32_t, or a low word with bit 31 set smears into the high half)
        const uint64_t u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32)) << 32) |
                           (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a);
        return reinterpret_cast<decltype(ptr)>(u);
    };
#ifndef VMV_ATTN_VGPR_PTRS      // (A/B hook: -DVMV_ATTN_VGPR_PTRS keeps the round-2 form)
    qp = uniform_ptr(qp); kp = uniform_ptr(kp); vp = uniform_ptr(vp); op = uniform_ptr(op);
#endif
    // row offsets inside one problem fit 32 bits (vmv_attention checks (N - 1) * s_row + head_dim < 2^30 elements): 64-bit
    // products of the row index kept their sign-extension registers alive across the main loop (three more spilled pairs)
    const int qs_row = (int)p.qm.s_row, ks_row = (int)p.km.s_row, vs_row = (int)p.vm.s_row, os_row = (int)p.om.s_row;

    // ---- staging group
    constexpr int NT = (WPP == 4) ? 256 : 64;
    const int stid = (WPP == 4) ? tid : lane;
    unsigned char* region = smem_raw + ((WPP == 4) ? 0 : wave * 8192);       // WPP = 1: a private V^T stage per wave
    // per stage: K tile [64 keys][8 slots] in 16-B units (8 KB) then V^T tile [64 d][64 keys] bf16 (8 KB)

    // ---- Q fragments (B operand): row q0 + 16*qt + u, k-slot kk*32 + 8*g
    elem8_t qf[QT][KK];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int q = q0 + qt * 16 + u;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (wvalid && q < p.Nq) v = *reinterpret_cast<const u32x4_t*>(qp + q * qs_row + kk * 32 + g * 8);
            qf[qt][kk] = __builtin_bit_cast(elem8_t, v);
        }
    }

    f32x4_t oacc[QT][DT];
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        m_run[qt] = NEG_BIG; l_run[qt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) oacc[qt][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    const float sc = p.scale * 1.44269504088896341f;   // exp2 domain (> 0)

    const int ntile = (p.Nk + 63) >> 6;
    // K/V staging.  WPP = 4: two LDS stages; the next tile's global loads are issued BEFORE the current tile's MFMAs
    // and written to the other stage after them (one barrier per tile, HBM/L2 latency hidden under the math).
    // WPP = 1 (one short problem per wave, a single tile in practice): plain load -> write -> barrier.
    // K / V tiles (WPP = 4) go global -> LDS by DMA (round 3; was global -> registers -> ds_write, 21 % of the kernel by ablation):
    // a wave-instruction fills 1 KB of the stage lane-linearly, so the lane FETCHES the piece whose place it writes — the swizzled
    // K image and the transpose-read V image below are both permutations of 16-byte pieces.  Rows >= Nk fall outside the
    // descriptors and arrive as zeros (0 x NaN from a recycled buffer would poison P.V).
    constexpr int NPIECE = (64 * SLOTS) / NT;      // 16-byte pieces of K (and of V) per lane per tile
    uint32_t k_off[(WPP == 4) ? NPIECE : 1], v_off[(WPP == 4) ? NPIECE : 1];
    __amdgpu_buffer_rsrc_t k_rsrc, v_rsrc;
    if constexpr (WPP == 4) {
        k_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(kp), 0, (uint32_t)((p.Nk - 1) * ks_row + D) * 2u, vmvg::SRD_FLAGS);
        v_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(vp), 0, (uint32_t)((p.Nk - 1) * vs_row + D) * 2u, vmvg::SRD_FLAGS);
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) {
            const int q = (wave * NPIECE + i) * 64 + lane;                   // piece of the tile image this lane fills
            const int krow = q >> SLOG, ksl = (q & (SLOTS - 1)) ^ (k_swz(krow) & (SLOTS - 1));
            k_off[i] = (uint32_t)(krow * ks_row + ksl * 8) * 2u;
            const int dt = q >> 7, r = q & 127;                              // V image: 2 KB (128 pieces) per 16-d subtile
            const int pos = (r >> 3) ^ (dt & 1);
            const int vrow = (((pos >> 3) & 1) << 5) | ((pos & 3) << 3) | (((pos >> 2) & 1) << 2) | ((r >> 1) & 3);
            v_off[i] = (uint32_t)(vrow * vs_row + (dt * 2 + (r & 1)) * 8) * 2u;
        }
    }
    auto issue_tile = [&](int kt2, int buf) {
        if constexpr (WPP == 4) {
            unsigned char* kd = region + buf * STAGE + wave * (NPIECE * 1024);
            const uint32_t kb = (uint32_t)(kt2 * 64 * ks_row) * 2u, vb = (uint32_t)(kt2 * 64 * vs_row) * 2u;
#pragma unroll
            for (int i = 0; i < NPIECE; ++i) {
                vmvg::blds16(k_rsrc, kd + i * 1024, k_off[i] + kb, 0);
                vmvg::blds16(v_rsrc, kd + KBYTES + i * 1024, v_off[i] + vb, 0);
            }
        }
    };
    auto write_vt = [&](uint16_t* Vtd, int idx, const u32x4_t& vv) {
        const int row = idx >> SLOG, slot = idx & (SLOTS - 1);
        const uint32_t w[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = slot * 8 + j;
            const int fk = ((slot ^ (((slot & 1) << 2) | (j >> 1))) & 7) << 3;   // = 8*(((d>>3)^(d>>1))&7)
            const uint16_t val = (j & 1) ? (uint16_t)(w[j >> 1] >> 16) : (uint16_t)(w[j >> 1] & 0xffffu);
            Vtd[d * 64 + (row ^ fk)] = val;
        }
    };
    // V (WPP = 4) stays ROW-major in LDS — one 16-byte piece per (key, 8 d) instead of eight scattered 2-byte writes of a
    // transposed image — and the PV fragments come from gfx950's transpose read: ds_read_b64_tr_b16 hands lane j of a
    // 16-lane group column j of the [4 rows][16 columns] block whose row j / 4, columns 4 (j % 4)..+3 the lanes address
    // (tools/experiments/tr_probe.hip).  Image: per 16-d subtile (2 KB) sixteen 128-byte blocks of [4 keys][16 d]; the
    // block of keys 32 kk + 8 g + 4 h + 0..3 sits at position (8 kk + 4 h + g) ^ (dt & 1), so that the two lane groups
    // of a half-wave (g, g + 1) read different 128-byte bank halves: piece (key, slot) at dt * 2048 + (pos << 7) + ((key & 3) << 5)
    // + ((slot & 1) << 4), dt = slot >> 1, pos = (((key >> 5) << 3) + (((key >> 2) & 1) << 2) + ((key >> 3) & 3)) ^ (dt & 1)
    // (issue_tile above inverts this).  K image: [key][slot ^ swizzle(key)].
    if constexpr (WPP == 4) {
        issue_tile(0, 0);
        vmvg::wait_vmcnt<0>();
        if constexpr (NST > 2) { if (ntile > 1) issue_tile(1, 1); }
        __syncthreads();
    }
    for (int kt = 0; kt < ntile; ++kt) {
        const int key0 = kt * 64;
        const u32x4_t* Ks;
        const uint16_t* Vt;
        if constexpr (WPP == 4) {
#if VMV_ATTN_ABLATE == 4 || VMV_ATTN_ABLATE == 5
            Ks = reinterpret_cast<const u32x4_t*>(region);
            Vt = reinterpret_cast<const uint16_t*>(region + KBYTES);
#else
            if constexpr (NST == 2) { if (kt + 1 < ntile) issue_tile(kt + 1, (kt + 1) & 1); }      // the other stage: last read one tile ago, behind a barrier
            Ks = reinterpret_cast<const u32x4_t*>(region + (kt % NST) * STAGE);
            Vt = reinterpret_cast<const uint16_t*>(region + (kt % NST) * STAGE + KBYTES);
#endif
        } else {
            // One short problem per wave (the 24-frame temporal attention: HBM-bound).  Only V^T goes through LDS — the K
            // fragments are 16 contiguous bytes of a key row and are loaded straight into registers below — so a wave
            // needs 8 KB instead of 16 (5 blocks per CU instead of 2), and the stage is private: no block barriers, the
            // wave's own LDS instructions execute in order.
            uint16_t* Vtd = reinterpret_cast<uint16_t*>(region);
            __builtin_amdgcn_s_waitcnt(0xc07f);   // previous tile's fragment reads are done
            asm volatile("" ::: "memory");
            u32x4_t vvr[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = stid + i * NT;
                const int key = key0 + (idx >> 3);
                vvr[i] = u32x4_t{0u, 0u, 0u, 0u};
                if (wvalid && key < p.Nk) vvr[i] = *reinterpret_cast<const u32x4_t*>(vp + key * vs_row + (idx & 7) * 8);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) write_vt(Vtd, stid + i * NT, vvr[i]);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            asm volatile("" ::: "memory");
            Ks = nullptr; Vt = Vtd;
        }

        // ---- S^T tiles (two query tiles at a time: the K fragments stay in registers for all QT of them) + online softmax
        elem8_t kf[4][KK];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int krow = 32 * (t >> 1) + 8 * (u >> 2) + 4 * (t & 1) + (u & 3);
            const int ksw = k_swz(krow);
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                if constexpr (WPP == 4) {
                    kf[t][kk] = __builtin_bit_cast(elem8_t, Ks[krow * SLOTS + ((kk * 4 + g) ^ (ksw & (SLOTS - 1)))]);
                } else {
                    u32x4_t kv = {0u, 0u, 0u, 0u};
                    const int key = key0 + krow;
                    if (wvalid && key < p.Nk) kv = *reinterpret_cast<const u32x4_t*>(kp + key * ks_row + kk * 32 + g * 8);
                    kf[t][kk] = __builtin_bit_cast(elem8_t, kv);
                }
            }
        }
        elem8_t pf[QT][2];
        const bool partial = key0 + 64 > p.Nk;                 // uniform
#pragma unroll
        for (int qp = 0; qp < QT / 2; ++qp) {
            f32x4_t s[2][4];
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                for (int t = 0; t < 4; ++t) s[q2][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                    for (int q2 = 0; q2 < 2; ++q2)
                        s[q2][t] = VMV_MFMA16(kf[t][kk], qf[2 * qp + q2][kk], s[q2][t], 0, 0, 0);
            // ---- online softmax (per lane: query u, 16 keys of this tile).  The softmax VALU work, not the 32 MFMAs, bounds a
            //      tile (16 quarter-rate v_exp per query tile alone cost as much as the tile's MFMAs), so it is kept minimal:
            //      the running max is tracked on the RAW scores (scale > 0) and the scale is folded into one FMA per score,
            //      exp2((s - m) * sc) = exp2(fma(s, sc, -m * sc)); key masking runs only on a partial last tile; the
            //      accumulator rescale is skipped while no lane of the wave sees a new maximum.
    #pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
                const int qt = 2 * qp + q2;
                if (partial || CAUSAL) {         // (causal: key <= query — the CLIP text tower; key 0 is visible to every query, so the
                                                 //  running max is real from the first tile on and masked scores exponentiate to 0)
                    const int qlim = CAUSAL ? q0 + qt * 16 + u : 0x7fffffff;
    #pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int kb = key0 + 32 * (t >> 1) + 8 * g + 4 * (t & 1);
    #pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (kb + r >= p.Nk || kb + r > qlim) s[q2][t][r] = NEG_BIG;
                    }
                }
                // (v_max3 through asm: fmaxf() is llvm.maxnum, which in IEEE mode first quiets every MFMA result with a
                //  v_max x, x — 12 extra instructions per query tile; the cross-lane steps are gfx950's VALU lane swaps
                //  instead of two ds_bpermute round trips)
#if VMV_ATTN_ABLATE == 2
                const float m_old = m_run[qt];
                const float m_new = m_old;
#else
                float mx = vmax3(s[q2][0][0], s[q2][0][1], s[q2][0][2]);
                mx = vmax3(mx, s[q2][0][3], s[q2][1][0]);
                mx = vmax3(mx, s[q2][1][1], s[q2][1][2]);
                mx = vmax3(mx, s[q2][1][3], s[q2][2][0]);
                mx = vmax3(mx, s[q2][2][1], s[q2][2][2]);
                mx = vmax3(mx, s[q2][2][3], s[q2][3][0]);
                mx = vmax3(mx, s[q2][3][1], s[q2][3][2]);
                mx = vmax2(mx, s[q2][3][3]);
                mx = xor16_max(mx);
                const float m_old = m_run[qt];
                const float m_new = xor32_max3(mx, m_old);
#endif
                const float nm = -m_new * sc;
                const f32x2_t sc2 = {sc, sc}, nm2 = {nm, nm};
                f32x2_t ps2 = {0.f, 0.f};
                float pv[4][4];
    #pragma unroll
                for (int t = 0; t < 4; ++t) {           // packed fp32: one v_pk_fma / v_pk_add per two scores
                    f32x2_t a = {s[q2][t][0], s[q2][t][1]}, b = {s[q2][t][2], s[q2][t][3]};
                    a = a * sc2 + nm2; b = b * sc2 + nm2;
#if VMV_ATTN_ABLATE == 1 || VMV_ATTN_ABLATE == 2
                    const f32x2_t ea = a, eb = b;
#else
                    const f32x2_t ea = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
                    const f32x2_t eb = {__builtin_amdgcn_exp2f(b.x), __builtin_amdgcn_exp2f(b.y)};
#endif
                    pv[t][0] = ea.x; pv[t][1] = ea.y; pv[t][2] = eb.x; pv[t][3] = eb.y;
                    ps2 += ea; ps2 += eb;
                }
                const float psum = ps2.x + ps2.y;
                if (__builtin_amdgcn_ballot_w64(m_new != m_old) != 0ull) {          // some query of this wave has a new maximum
                    const float alpha = __builtin_amdgcn_exp2f((m_old - m_new) * sc);
                    m_run[qt] = m_new;
                    l_run[qt] *= alpha;
    #pragma unroll
                    for (int dt = 0; dt < DT; ++dt) oacc[qt][dt] *= alpha;
                }
                l_run[qt] += psum;
    #pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    u32x4_t w;
                    w.x = pack_elem2(pv[2 * kk][0], pv[2 * kk][1]);
                    w.y = pack_elem2(pv[2 * kk][2], pv[2 * kk][3]);
                    w.z = pack_elem2(pv[2 * kk + 1][0], pv[2 * kk + 1][1]);
                    w.w = pack_elem2(pv[2 * kk + 1][2], pv[2 * kk + 1][3]);
                    pf[qt][kk] = __builtin_bit_cast(elem8_t, w);
                }
            }
        }
        // ---- O^T += V^T P^T
        if constexpr (WPP == 4 && NST > 2) {     // stage (kt + 2) % 3 was last read in tile kt - 1, behind that tile's barrier
            if (kt + 2 < ntile) issue_tile(kt + 2, (kt + 2) % NST);
        }
        if constexpr (WPP == 4) {
            typedef short s16x4_t __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_p;
            const unsigned char* vlane = reinterpret_cast<const unsigned char*>(Vt) + ((u >> 2) << 5) + ((u & 3) << 3);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
#if defined(__HIP_DEVICE_COMPILE__)
                    const unsigned char* b0 = vlane + dt * 2048 + (((kk * 8 + g) ^ (dt & 1)) << 7);
                    const unsigned char* b1 = vlane + dt * 2048 + (((kk * 8 + 4 + g) ^ (dt & 1)) << 7);
                    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(b0));
                    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(b1));
                    const u32x2_t lo2 = __builtin_bit_cast(u32x2_t, lo), hi2 = __builtin_bit_cast(u32x2_t, hi);
                    const elem8_t vf = __builtin_bit_cast(elem8_t, u32x4_t{lo2.x, lo2.y, hi2.x, hi2.y});
#else
                    const elem8_t vf = {};
#endif
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) {
#if VMV_ATTN_ABLATE == 3
                        if (kk == 0 && dt == 0) { oacc[qt][0][0] += (float)vf[0] + (float)pf[qt][0][0] + (float)pf[qt][1][0]; }
#else
                        oacc[qt][dt] = VMV_MFMA16(vf, pf[qt][kk], oacc[qt][dt], 0, 0, 0);
#endif
                    }
                }
            }
        } else {
            const u32x4_t* Vt16 = reinterpret_cast<const u32x4_t*>(Vt);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int d = dt * 16 + u;
                const int fsl = ((d >> 3) ^ (d >> 1)) & 7;   // 8-key block swizzle of row d
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const elem8_t vf = __builtin_bit_cast(elem8_t, Vt16[d * 8 + ((kk * 4 + g) ^ fsl)]);
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt)
                        oacc[qt][dt] = VMV_MFMA16(vf, pf[qt][kk], oacc[qt][dt], 0, 0, 0);
                }
            }
        }
        if constexpr (WPP == 4) {
#if VMV_ATTN_ABLATE == 4
            __syncthreads();
#elif VMV_ATTN_ABLATE == 5
#else
            // my pieces of tile kt + 1 have landed (three stages: those of tile kt + 2, issued above, may still be in flight) ...
            if (NST > 2 && kt + 2 < ntile) vmvg::wait_vmcnt<2 * NPIECE>(); else vmvg::wait_vmcnt<0>();
            __syncthreads();                                   // ... and everyone's; everyone is done reading tile kt
#endif
        }
    }

    // ---- normalise and store: lane owns O[q = q0 + 16 qt + u][d = 16 dt + 4 g + 0..3]
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const float l = xor16_32_sum(l_run[qt]);
        const float inv = 1.0f / l;
        const int q = q0 + qt * 16 + u;
        if (wvalid && q < p.Nq) {
            uint16_t* orow = op + q * os_row + 4 * g;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const f32x4_t a = oacc[qt][dt] * inv;
                u32x2_t w;
                w.x = pack_elem2(a.x, a.y); w.y = pack_elem2(a.z, a.w);
                *reinterpret_cast<u32x2_t*>(orow + dt * 16) = w;
            }
        }
    }
}

// ---- short problems (Nq, Nk <= 32: the 24-frame temporal attention of every pixel x head) — HBM-bound.
// One wave per problem, as attn_kernel<1, .>, but built for memory-level parallelism instead of generality:
//   * every global load of the problem (Q: 2, K: 4, V: 4 wave-instructions of 16 B per lane) is issued before the first
//     use, so a wave pays ONE memory round trip instead of three dependent ones;
//   * only the 32-key half that can hold valid keys exists: 2 S^T tiles instead of 4, one P.V k-step instead of 2 (half
//     the MFMAs, the exponentials and the registers: 4-5 waves per SIMD instead of 2-3), and the V^T stage is
//     [64 d][32 keys] = 4 KB per wave, filled from the 4 valid row groups only (32 instead of 64 16-bit LDS writes);
//   * the stage is private to the wave: no block barrier anywhere.
// Same transposed formulation and key permutation as attn_kernel (header of this file).
__global__ __launch_bounds__(256, 4) void attn_short_kernel(const VmvAttnParams p, const int nproblems) {
    VMV_KERNEL_ENTER();
    __shared__ __attribute__((aligned(16))) unsigned char smem_s[4 * 4096];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int u = lane & 15, g = lane >> 4;
    const int pidx = blockIdx.x * 4 + wave;
    if (pidx >= nproblems) return;                      // (no block-level synchronisation below)
    const int h = pidx % p.heads, o = pidx / p.heads;
    const uint16_t* qp = reinterpret_cast<const uint16_t*>(p.q) + seq_base(p.qm, o) + h * 64;
    const uint16_t* kp = reinterpret_cast<const uint16_t*>(p.k) + seq_base(p.km, o / p.kv_div) + h * 64;
    const uint16_t* vp = reinterpret_cast<const uint16_t*>(p.v) + seq_base(p.vm, o / p.kv_div) + h * 64;
    uint16_t* op = reinterpret_cast<uint16_t*>(p.o) + seq_base(p.om, o) + h * 64;
    uint16_t* Vtd = reinterpret_cast<uint16_t*>(smem_s + wave * 4096);

    // ---- all loads of the problem, back to back
    u32x4_t qv[2][2], kv[2][2], vv[4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = qt * 16 + u;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            qv[qt][kk] = u32x4_t{0u, 0u, 0u, 0u};
            if (q < p.Nq) qv[qt][kk] = *reinterpret_cast<const u32x4_t*>(qp + (long)q * p.qm.s_row + kk * 32 + g * 8);
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int key = 8 * (u >> 2) + 4 * t + (u & 3);          // permuted key order (file header), keys 0..31
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            kv[t][kk] = u32x4_t{0u, 0u, 0u, 0u};
            if (key < p.Nk) kv[t][kk] = *reinterpret_cast<const u32x4_t*>(kp + (long)key * p.km.s_row + kk * 32 + g * 8);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int key = (lane >> 3) + 8 * i;
        vv[i] = u32x4_t{0u, 0u, 0u, 0u};
        if (key < p.Nk) vv[i] = *reinterpret_cast<const u32x4_t*>(vp + (long)key * p.vm.s_row + (lane & 7) * 8);
    }
    // ---- V^T stage: element (d, key) at Vtd[d * 32 + (key ^ 8 * (((d >> 3) ^ (d >> 1)) & 3))]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int key = (lane >> 3) + 8 * i, slot = lane & 7;
        const uint32_t w[4] = {vv[i].x, vv[i].y, vv[i].z, vv[i].w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = slot * 8 + j;
            const int fk = ((slot ^ (j >> 1)) & 3) << 3;
            Vtd[d * 32 + (key ^ fk)] = (j & 1) ? (uint16_t)(w[j >> 1] >> 16) : (uint16_t)(w[j >> 1] & 0xffffu);
        }
    }
    // ---- S^T = K Q^T (2 key tiles x 2 query tiles) and the softmax of each query over its <= 32 keys
    const float sc = p.scale * 1.44269504088896341f;
    elem8_t pf[2];
    float linv[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        f32x4_t s[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            s[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                s[t] = VMV_MFMA16(__builtin_bit_cast(elem8_t, kv[t][kk]), __builtin_bit_cast(elem8_t, qv[qt][kk]), s[t], 0, 0, 0);
        }
        // lane (u, g) holds the scores of query 16 qt + u against keys 8 g + 4 t + r
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (8 * g + 4 * t + r >= p.Nk) s[t][r] = NEG_BIG;
        float mx = vmax3(s[0][0], s[0][1], s[0][2]);
        mx = vmax3(mx, s[0][3], s[1][0]);
        mx = vmax3(mx, s[1][1], s[1][2]);
        mx = vmax2(mx, s[1][3]);
        mx = xor32_max3(xor16_max(mx), NEG_BIG);
        const float nm = -mx * sc;
        float e[2][4], psum = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                e[t][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][r], sc, nm));
                psum += e[t][r];
            }
        linv[qt] = 1.0f / xor16_32_sum(psum);
        u32x4_t w;
        w.x = pack_elem2(e[0][0], e[0][1]); w.y = pack_elem2(e[0][2], e[0][3]);
        w.z = pack_elem2(e[1][0], e[1][1]); w.w = pack_elem2(e[1][2], e[1][3]);
        pf[qt] = __builtin_bit_cast(elem8_t, w);               // = P^T B-operand for keys 8 g + 0..7
    }
    // ---- O^T = V^T P^T (the wave's own LDS writes above are complete: same-wave LDS ops execute in order)
    __builtin_amdgcn_s_waitcnt(0xc07f);
    asm volatile("" ::: "memory");
    const u32x4_t* Vt16 = reinterpret_cast<const u32x4_t*>(Vtd);
    f32x4_t oacc[2][4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const int d = dt * 16 + u;
        const int fsl = ((d >> 3) ^ (d >> 1)) & 3;
        const elem8_t vf = __builtin_bit_cast(elem8_t, Vt16[d * 4 + (g ^ fsl)]);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
            oacc[qt][dt] = VMV_MFMA16(vf, pf[qt], f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = qt * 16 + u;
        if (q < p.Nq) {
            uint16_t* orow = op + (long)q * p.om.s_row + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const f32x4_t a = oacc[qt][dt] * linv[qt];
                u32x2_t w;
                w.x = pack_elem2(a.x, a.y); w.y = pack_elem2(a.z, a.w);
                *reinterpret_cast<u32x2_t*>(orow + dt * 16) = w;
            }
        }
    }
}

int map_ok(const VmvSeqMap& m) {
    return m.inner > 0 && !(m.s_outer & 7) && !(m.s_inner & 7) && !(m.s_row & 3);
}

}  // namespace

extern "C" int vmv_attention(const VmvAttnParams* pp, void* stream) {
    if (!pp) return VMV_ENULL;
    const VmvAttnParams& p = *pp;
    if (!p.q || !p.k || !p.v || !p.o) return VMV_ENULL;
    if (p.n_outer <= 0 || p.heads <= 0 || p.Nq <= 0 || p.Nk <= 0 || p.kv_div <= 0) return VMV_EINVAL;
    {   // the kernels index the rows of one problem with 32-bit element offsets; the 4-waves-per-problem kernels issue K / V DMA
        // offsets for the whole last 64-row tile (rows >= Nk must land OUTSIDE the descriptor, i.e. must not wrap), and a row is
        // head_dim elements wide: validate the PADDED counts with the actual head_dim
        const long lim = 1L << 30;
        const long hdw = p.head_dim ? p.head_dim : 64;
        const long nq_pad = ((long)p.Nq + 255) / 256 * 256, nk_pad = ((long)p.Nk + 63) / 64 * 64;
        if ((nq_pad - 1) * p.qm.s_row + hdw >= lim || (nq_pad - 1) * p.om.s_row + hdw >= lim || (nk_pad - 1) * p.km.s_row + hdw >= lim ||
            (nk_pad - 1) * p.vm.s_row + hdw >= lim || p.qm.s_row < 0 || p.km.s_row < 0 || p.vm.s_row < 0 || p.om.s_row < 0) return VMV_ERANGE;
    }
    if (!map_ok(p.qm) || !map_ok(p.km) || !map_ok(p.vm) || !map_ok(p.om)) return VMV_EALIGN;
    if ((p.qm.s_row & 7) || (p.km.s_row & 7) || (p.vm.s_row & 7)) return VMV_EALIGN;
    if (!vmv_aligned16(p.q) || !vmv_aligned16(p.k) || !vmv_aligned16(p.v) || (((uintptr_t)p.o) & 7)) return VMV_EALIGN;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int hd = p.head_dim ? p.head_dim : 64;
    if (p.causal && (hd != 64 || p.Nq != p.Nk)) return VMV_EINVAL;          // causal: self-attention on the general head_dim-64 kernel
    if (hd == 32) {                  // LGM MVAttention (core/attention.py:67-84): long sequences only
        if (p.n_outer > 65535 || p.heads > 65535) return VMV_ERANGE;
        hipLaunchKernelGGL((attn_kernel<4, 2, 32>), dim3((p.Nq + 127) / 128, p.heads, p.n_outer), dim3(256), VMV_ATTN_STAGES * 16384, st, p, 0);
        return vmv_launch_status();
    }
    if (hd == 128) {                 // zero-padded wide heads (the CLIP image tower's head_dim 80 packed to 128: clip_vision.py)
        if (p.n_outer > 65535 || p.heads > 65535) return VMV_ERANGE;
        static std::atomic<unsigned long long> attr128{0};
        if (const int rc_attr = vmv_lds_attr_once(attr128, reinterpret_cast<const void*>(&attn_kernel<4, 2, 128>), 65536)) return rc_attr;
        hipLaunchKernelGGL((attn_kernel<4, 2, 128>), dim3((p.Nq + 127) / 128, p.heads, p.n_outer), dim3(256), 65536, st, p, 0);
        return vmv_launch_status();
    }
    if (hd != 64) return VMV_EINVAL;
    static int short_env = -1;
    if (short_env < 0) { const char* e = getenv("VMV_ATTN_SHORT"); short_env = e ? atoi(e) : 1; }
    if (p.Nq <= 32 && p.Nk <= 32 && short_env && !p.causal) {
        const int nproblems = p.n_outer * p.heads;
        hipLaunchKernelGGL(attn_short_kernel, dim3((nproblems + 3) / 4), dim3(256), 0, st, p, nproblems);
    } else if (p.Nq <= 32 && !p.causal) {
        const int nproblems = p.n_outer * p.heads;
        static std::atomic<unsigned long long> attr1{0};
        if (const int rc_attr = vmv_lds_attr_once(attr1, reinterpret_cast<const void*>(&attn_kernel<1, 2>), 32768)) return rc_attr;
        hipLaunchKernelGGL((attn_kernel<1, 2>), dim3((nproblems + 3) / 4), dim3(256), 32768, st, p, nproblems);
    } else {
        if (p.n_outer > 65535 || p.heads > 65535) return VMV_ERANGE;
        static int qt_env = -1;
        if (qt_env < 0) { const char* e = getenv("VMV_ATTN_QT"); qt_env = e ? atoi(e) : 0; }
        // (round 6: 64-query blocks — QT = 1, 84 registers, four blocks per CU — for the short problems (Nk <= 192 or Nq <= 192: cross-
        //  attention, third-level / middle self-attention) measured step-neutral: 47.88 / 47.91 vs 47.86 / 47.87 ms, attention family
        //  4.16 / 4.13 vs 4.17 / 4.17 ms, profiles/r6_attn_q64_step_ab.log — not kept)
        // 256-query blocks when that still leaves >= 2 blocks per CU and the key loop is long enough to matter
        const long blocks256 = (long)((p.Nq + 255) / 256) * p.heads * p.n_outer;
        const bool big = qt_env == 4 || (qt_env == 0 && p.Nk >= 512 && blocks256 >= 512 && (p.Nq % 256 == 0 || p.Nq >= 2048));
        if (p.causal) hipLaunchKernelGGL((attn_kernel<4, 2, 64, true>), dim3((p.Nq + 127) / 128, p.heads, p.n_outer), dim3(256), VMV_ATTN_STAGES * 16384, st, p, 0);
        else if (big) hipLaunchKernelGGL((attn_kernel<4, 4>), dim3((p.Nq + 255) / 256, p.heads, p.n_outer), dim3(256), VMV_ATTN_STAGES * 16384, st, p, 0);
        else hipLaunchKernelGGL((attn_kernel<4, 2>), dim3((p.Nq + 127) / 128, p.heads, p.n_outer), dim3(256), VMV_ATTN_STAGES * 16384, st, p, 0);
    }
    return vmv_launch_status();
}

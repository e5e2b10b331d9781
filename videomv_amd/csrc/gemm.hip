// gemm.hip — bf16 MFMA implicit-GEMM for gfx950 (MI355X): linear / 1x1 / 3x3 (s1, s2, nearest-x2) / temporal
// (3,1,1) convolutions with a fused epilogue.  See include/vmv.h for the contract and DESIGN.md §4.1.
//
// Mapping to the hardware
//   * block = 256 threads = 4 waves (2 x 2); wave tile = (16*WM) activation rows x (16*WN) output channels,
//     built from v_mfma_f32_16x16x32_bf16.  The MFMA "A" operand is the WEIGHT fragment and the "B" operand
//     the ACTIVATION fragment, i.e. we compute out^T: every lane then owns 4 consecutive output channels of
//     one activation row, so the epilogue loads/stores 8-byte (bf16x4) / 16-byte (fp32x4) vectors.
//   * K is walked in 64-channel chunks over a list of (source, tap) segments; both operand tiles are staged
//     global -> registers -> LDS (the activation gather needs zero-fill predication, so LDS-DMA is not used),
//     double buffered with ONE barrier per chunk: loads for chunk t+1 are issued before the MFMAs of chunk t
//     and written to the other LDS buffer after them.
//   * LDS tiles are [rows][64] bf16 (128-B rows) with the 16-B slot index XOR-swizzled by (row>>1)&7 so that
//     ds_read_b128 fragment reads and ds_write_b128 staging writes are bank-conflict free.
//   * blockIdx -> tile is XCD-aware (8 XCDs, private L2): each XCD gets a contiguous range of tiles and walks
//     the N tiles of one M tile first, so the activation tile is re-used out of that XCD's L2.
#include "gemm_common.h"
#include <cstdlib>

using namespace vmvg;

int vmv_gemm_glds_launch(const VmvGemmParams& p, int total_steps, int tile, hipStream_t st);   // gemm_glds.hip
int vmv_gemm_pglds_launch(const VmvGemmParams& p, int total_steps, int tile, hipStream_t st);  // gemm_pglds.hip
bool vmv_gemm_pglds_supported(const VmvGemmParams& p);
int vmv_gemm_xglds_launch(const VmvGemmParams& p, int total_steps, int tile, hipStream_t st);  // gemm_xglds.hip
int vmv_gemm_xglds_epi_ok(const VmvGemmParams& p, int tile);                                      // gemm_xglds.hip
#if defined(VMV_EXPERIMENTS)
int vmv_gemm_sglds_launch(const VmvGemmParams& p, int total_steps, int tile, hipStream_t st);  // gemm_sglds.hip
int vmv_gemm_astat_launch(const VmvGemmParams& p, int tile, hipStream_t st);                   // gemm_astat.hip
int vmv_gemm_wreg_launch(const VmvGemmParams& p, hipStream_t st);                             // gemm_wreg.hip (round 6)
bool vmv_gemm_wreg_supported(const VmvGemmParams& p);
#else      // production build: the measured-and-rejected kernels are not in the library (make EXPERIMENTS=1)
constexpr int VMV_NOT_BUILT = -101;
static int vmv_gemm_sglds_launch(const VmvGemmParams&, int, int, hipStream_t) { return VMV_NOT_BUILT; }
static int vmv_gemm_astat_launch(const VmvGemmParams&, int, hipStream_t) { return VMV_NOT_BUILT; }
static int vmv_gemm_wreg_launch(const VmvGemmParams&, hipStream_t) { return VMV_NOT_BUILT; }
static bool vmv_gemm_wreg_supported(const VmvGemmParams&) { return false; }
#endif
int vmv_gemm_rs_launch(const VmvGemmParams& p, int tile, hipStream_t st);                      // gemm_rs.hip
bool vmv_gemm_rs_supported(const VmvGemmParams& p);
int vmv_conv_halo_launch(const VmvGemmParams& p, hipStream_t st);                             // conv_halo.hip
bool vmv_conv_halo_supported(const VmvGemmParams& p);
bool vmv_gemm_rs_preferred(const VmvGemmParams& p);
int vmv_gemm_tfr_launch(const VmvGemmParams& p, hipStream_t st);                              // gemm_tfr.hip
bool vmv_gemm_tfr_supported(const VmvGemmParams& p);
bool vmv_gemm_tfr_preferred(const VmvGemmParams& p);
int vmv_gemm_tqa_launch(const VmvGemmParams& p, hipStream_t st);                              // gemm_tqa.hip
bool vmv_gemm_tqa_supported(const VmvGemmParams& p);
bool vmv_gemm_tqa_preferred(const VmvGemmParams& p);
#if defined(VMV_EXPERIMENTS)
bool vmv_gemm_astat_eligible(const VmvGemmParams& p);
#else
static bool vmv_gemm_astat_eligible(const VmvGemmParams&) { return false; }
#endif

namespace {

template <int WM, int WN>
struct GemmCfg {
    static constexpr int BM = 32 * WM;
    static constexpr int BN = 32 * WN;
    static constexpr int LDS_BYTES = 2 * (BM + BN) * BK * 2;
};

template <int WM, int WN>
__global__ __launch_bounds__(256) void gemm_kernel(const VmvGemmParams p, const int tiles_m, const int tiles_n,
                                                   const int total_steps, const int steps_per_split) {
    VMV_KERNEL_ENTER();
    using Cfg = GemmCfg<WM, WN>;
    constexpr int BM = Cfg::BM, BN = Cfg::BN;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4_t* const lds = reinterpret_cast<u32x4_t*>(smem_raw);
    // layout (in 16-B units): A buf0 [BM*8] | A buf1 | W buf0 [BN*8] | W buf1
    constexpr int A_U = BM * 8, W_U = BN * 8;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_m = wave >> 1, wave_n = wave & 1;

    // ---- XCD-aware tile mapping (bijective for any block count)
    const int nblk = tiles_m * tiles_n;
    int logical;
    {
        const int bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = logical % tiles_n;
    const int tile_m = logical / tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int split = blockIdx.y;
    const int step_begin = split * steps_per_split;
    const int step_end = min(total_steps, step_begin + steps_per_split);

    // ---- staging assignment: thread -> (row srow + 32*i, 16-B slot sslot)
    const int srow = tid >> 3, sslot = tid & 7;
    const int sw_slot = sslot ^ ((srow >> 1) & 7);

    RowInfo rinfo[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int m = m0 + srow + 32 * i;
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
        rinfo[i] = r;
    }
    const uint16_t* wrow[WN];
    bool wvalid[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int n = n0 + srow + 32 * j;
        wvalid[j] = n < p.N;
        wrow[j] = reinterpret_cast<const uint16_t*>(p.W) + (size_t)(wvalid[j] ? n : 0) * p.ktot + sslot * 8 +
                  (p.wgroup_rows > 0 ? (long)(m0 / p.wgroup_rows) * p.wgroup_stride : 0L);        // grouped weights (vmv.h)
    }

    // ---- K-walk state (segment s, chunk offset kc inside it, cumulative weight offset koff)
    int s = 0, kc = 0, koff = 0;
    {
        int skip = step_begin;
        while (s < p.nseg) {
            const int nch = (p.seg[s].k + BK - 1) / BK;
            if (skip < nch) { kc = skip * BK; break; }
            skip -= nch; koff += p.seg[s].k; ++s;
        }
    }
    int aoff[WM];
    auto enter_segment = [&]() {
#pragma unroll
        for (int i = 0; i < WM; ++i) aoff[i] = seg_row_offset(p, p.seg[s], rinfo[i]);
    };
    if (s < p.nseg) enter_segment();

    u32x4_t ra[WM], rw[WN];
    auto load_chunk = [&]() {
        const VmvGemmSeg& sg = p.seg[s];
        const int kk = kc + sslot * 8;
        const bool kvalid = kk < sg.k;
        const uint16_t* src = reinterpret_cast<const uint16_t*>(sg.src) + kk;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (kvalid && aoff[i] >= 0) v = *reinterpret_cast<const u32x4_t*>(src + aoff[i]);
            ra[i] = v;
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (kvalid && wvalid[j]) v = *reinterpret_cast<const u32x4_t*>(wrow[j] + koff + kc);
            rw[j] = v;
        }
    };
    auto advance = [&]() {
        kc += BK;
        if (kc >= p.seg[s].k) {
            koff += p.seg[s].k; ++s; kc = 0;
            if (s < p.nseg) enter_segment();
        }
    };
    auto store_chunk = [&](int buf) {
        u32x4_t* a = lds + buf * A_U;
        u32x4_t* w = lds + 2 * A_U + buf * W_U;
#pragma unroll
        for (int i = 0; i < WM; ++i) a[(srow + 32 * i) * 8 + sw_slot] = ra[i];
#pragma unroll
        for (int j = 0; j < WN; ++j) w[(srow + 32 * j) * 8 + sw_slot] = rw[j];
    };

    f32x4_t acc[WN][WM];
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int i = 0; i < WM; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15;           // row inside a 16-row fragment
    const int fgrp = lane >> 4;           // k-group 0..3 (8 bf16 each)
    const int fswz = (frow >> 1) & 7;

    const int nsteps = step_end - step_begin;
    if (nsteps > 0) {
        load_chunk();
        advance();
        store_chunk(0);
    }
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < nsteps; ++t) {
        const bool more = (t + 1) < nsteps;
        if (more) { load_chunk(); advance(); }
        const u32x4_t* a = lds + cur * A_U + (wave_m * 16 * WM + frow) * 8;
        const u32x4_t* w = lds + 2 * A_U + cur * W_U + (wave_n * 16 * WN + frow) * 8;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int slot = (kk * 4 + fgrp) ^ fswz;
            elem8_t af[WM], wf[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) af[i] = __builtin_bit_cast(elem8_t, a[i * 16 * 8 + slot]);
#pragma unroll
            for (int j = 0; j < WN; ++j) wf[j] = __builtin_bit_cast(elem8_t, w[j * 16 * 8 + slot]);
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int i = 0; i < WM; ++i)
                    acc[j][i] = VMV_MFMA16(wf[j], af[i], acc[j][i], 0, 0, 0);
        }
        if (more) store_chunk(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: lane owns row m = .. + (lane&15), channels n = .. + 4*(lane>>4) + {0..3}
    const int mbase = m0 + wave_m * 16 * WM + frow;
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
    if (p.epilogue == VMV_EPI_GEGLU) {
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
}

// split-K second pass: sum the fp32 slabs in a fixed order (deterministic) and run the epilogue.
__global__ __launch_bounds__(256) void gemm_splitk_reduce(const VmvGemmParams p) {
    VMV_KERNEL_ENTER();
    const int geglu = p.epilogue == VMV_EPI_GEGLU;
    const int nq = geglu ? (p.N / 32) * 4 : p.N / 4;       // work items per row
    const long total = (long)p.M * nq;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int m = (int)(idx / nq);
        const int qi = (int)(idx - (long)m * nq);
        const int n = geglu ? (qi >> 2) * 32 + (qi & 3) * 4 : qi * 4;
        f32x4_t v = {0.f, 0.f, 0.f, 0.f}, g = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < p.ksplit; ++s) {
            const float* ws = p.workspace + ((size_t)s * p.M + m) * p.N + n;
            v += *reinterpret_cast<const f32x4_t*>(ws);
            if (geglu) g += *reinterpret_cast<const f32x4_t*>(ws + 16);
        }
        epilogue_store(p, m, n, v, g);
    }
}

template <int WM, int WN>
int launch_cfg(const VmvGemmParams& p, int total_steps, hipStream_t st) {
    using Cfg = GemmCfg<WM, WN>;
    const int tiles_m = (p.M + Cfg::BM - 1) / Cfg::BM;
    const int tiles_n = (p.N + Cfg::BN - 1) / Cfg::BN;
    const int ks = p.ksplit > 1 ? p.ksplit : 1;
    const int sps = (total_steps + ks - 1) / ks;
    static std::atomic<unsigned long long> attr_set{0};
    if (const int rc_attr = vmv_lds_attr_once(attr_set, reinterpret_cast<const void*>(&gemm_kernel<WM, WN>), Cfg::LDS_BYTES)) return rc_attr;
    dim3 grid(tiles_m * tiles_n, ks, 1);
    VMV_LAUNCH((gemm_kernel<WM, WN>), grid, dim3(256), Cfg::LDS_BYTES, st, p, tiles_m, tiles_n, total_steps, sps);
    return vmv_launch_status();
}

int gemm_policy() {
    // VMV_GEMM_POLICY (A/B experiments): 0 = 128-row register-staged kernel only, 1 = + the 256-row LDS-DMA kernel,
    // 2 (default) = + the persistent LDS-DMA kernel for short-K linears.
    static int pol = -1;
    if (pol < 0) {
        const char* e = getenv("VMV_GEMM_POLICY");
        pol = e ? atoi(e) : 2;
    }
    return pol;
}

int astat_policy() {
    // VMV_GEMM_ASTAT (A/B experiments): 1 = the A-stationary deferred-epilogue kernel (gemm_astat.hip) takes the K <= 320,
    // wide-N linears of the largest level (LayerNorm-folded qkv / q, GEGLU); 0 (default) = off.  Measured (round 2, same
    // box): +12-25 % on the L0 qkv shapes in isolation, GEGLU +-2 %, but the full step gets 0.6 ms SLOWER — with every MFMA
    // removed (VMV_GEMM_ABLATE=1) the kernel is no faster, without its stores +15-33 %: like the tile-per-item kernels it is
    // bound by the CU's vector-memory path (DMA issue + in-order vmcnt behind store round trips), not by the matrix pipes.
    static int pol = -1;
    if (pol < 0) {
        const char* e = getenv("VMV_GEMM_ASTAT");
        pol = e ? atoi(e) : 0;
    }
    return pol;
}

// Step-level A/B hooks (experiments): VMV_GEMM_TILE_GEGLU / _LIN160 / _LIN128 force a tile id for the short-K (<= 24 chunks)
// LINEAR GEMMs with GEGLU / N % 160 == 0 / other N, M >= 16384 (the L0 / L1 transformer linears); 0 = policy below.
int tile_override(int which) {
    static int ov[3] = {-1, -1, -1};
    static const char* names[3] = {"VMV_GEMM_TILE_GEGLU", "VMV_GEMM_TILE_LIN160", "VMV_GEMM_TILE_LIN128"};
    if (ov[which] < 0) { const char* e = getenv(names[which]); ov[which] = e ? atoi(e) : 0; }
    return ov[which];
}

int x512_policy() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("VMV_GEMM_X512"); v = e ? atoi(e) : 1; }
    return v;
}

int xglds_policy() {
    // VMV_GEMM_XGLDS (A/B experiments): 1 (default) = the wide-tile kernel (gemm_xglds.hip: 256 x 320 tiles, 64 x 160 wave tiles,
    // four-stage ring of 32-deep chunks) takes the long-K GATHERED GEMMs — 3x3 convolutions, temporal convolutions — whose
    // tiles fill the chip (the two large levels); 0 = off.  Measured (round 2, one box): L0 conv 806 -> 953, L0 temporal conv
    // 683 -> 787, L1 conv 1040 -> 1140, L1 temporal conv 901 -> 980, VAE 512- / 256-channel convs +12 / +9 % (256 x 256 tiles) TFLOP/s;
    // the plain-row linears (FF down, K = 1280 / 2560) are 1-5 % faster on the persistent kernel and the 128-channel VAE level on
    // gemm_glds, and stay there; full step -0.9 ms.
    static int pol = -1;
    if (pol < 0) {
        const char* e = getenv("VMV_GEMM_XGLDS");
        pol = e ? atoi(e) : 1;
    }
    return pol;
}

int conv_halo_policy() {
    // VMV_CONV_HALO (A/B experiments): 1 (default) = the halo-resident kernel takes the eligible few-channel 3 x 3 convolutions
    static int pol = -1;
    if (pol < 0) { const char* e = getenv("VMV_CONV_HALO"); pol = e ? atoi(e) : 1; }
    return pol;
}

int pick_tile(const VmvGemmParams& p, int total_steps) {
    if (p.epilogue == VMV_EPI_TATTN) return VMV_TILE_TQA;      // fused q | k | v + temporal attention: one kernel (vmv_gemm checks eligibility)
    if (p.tile != VMV_TILE_AUTO) return p.tile;
    const int geglu = p.epilogue == VMV_EPI_GEGLU;
    // the short-K linears of the two large levels: rows resident in registers, W streamed, outputs per column pair (gemm_rs.hip)
    if (gemm_policy() >= 2 && vmv_gemm_rs_preferred(p)) return VMV_TILE_RS;
    // 3 x 3 convolutions with N <= 8 output channels (the VAE / UNet heads): halo tile + weights in LDS (conv_halo.hip)
    if (gemm_policy() >= 2 && conv_halo_policy() && vmv_conv_halo_supported(p)) return VMV_TILE_HALO;
    // temporal convolutions whose frame-resident tiles fill the chip (gemm_tfr.hip); a GroupNorm folded into a temporal convolution
    // lives in that kernel only
    if (gemm_policy() >= 2 && vmv_gemm_tfr_preferred(p)) return VMV_TILE_TFR;
    if (p.gn_table && p.nseg == 3 && p.seg[0].mode == VMV_SEG_TEMPORAL) return VMV_TILE_TFR;
    if (p.gn_table) return VMV_TILE_RS;             // a folded GroupNorm lives in that kernel's prologue only (vmv_gemm checks eligibility)
    if (gemm_policy() >= 2 && xglds_policy() && !geglu && p.ksplit <= 1 && !p.rowstat && !vmv_gemm_ln_inline(p) && total_steps >= 12 &&
        (p.N % 320 == 0 || p.N % 256 == 0)) {
        bool any_gather = false;
        for (int i = 0; i < p.nseg; ++i) any_gather = any_gather || p.seg[i].mode != VMV_SEG_LINEAR;
        // one 256 x 320 (256 x 256) tile costs about 2 / 1.1 tiles of the 256 x 160 (256 x 128) kernel: take it when its rounds over
        // the 256 CUs are no more than that many of the other's (the UNet's two large levels, the VAE's 256- / 512-channel
        // levels at 24 frames; at the UNet's third level 120 tiles would leave half the chip idle)
        const int bx = p.N % 320 == 0 ? 320 : 256;
        const long tm = (p.M + 255) / 256;
        const long rounds_x = (tm * (p.N / bx) + 255) / 256, rounds_g = (tm * (p.N / (bx / 2)) + 255) / 256;
        // (plain rows too once the reduction is long: FF-down of the second level, K = 2560 — 937 -> 1008 TFLOP/s)
        if ((any_gather || total_steps >= 32) && 20 * rounds_x <= 11 * rounds_g) return bx == 320 ? VMV_TILE_X256x320 : VMV_TILE_X256x256;
    }
    // Round 6: N = 128 convolutions over >= 2 rounds of 512-row tiles (the VAE's first level: 1.6-3.9 M rows x 128 channels) — every
    // kernel with 64 x 64 wave tiles runs them at ~670 TFLOP/s (0.5 fragment reads per MFMA); the 8 x 1 wave grid's 64 x 128 wave tiles
    // (gemm_xglds.hip WNV = 1) read 0.375.  VMV_GEMM_X512=0 keeps the old choice (A/B).
    if (gemm_policy() >= 2 && xglds_policy() && x512_policy() && !geglu && p.ksplit <= 1 && !p.rowstat && !vmv_gemm_ln_inline(p) && p.N == 128 &&
        total_steps >= 12 && p.wgroup_rows == 0 && !p.out_fp32 && (long)p.M >= 2L * 256 * 512) {
        bool any_gather = false;
        for (int i = 0; i < p.nseg; ++i) any_gather = any_gather || p.seg[i].mode != VMV_SEG_LINEAR;
        if (any_gather) return VMV_TILE_X512x128;
    }
    // Round 4: the transformer linears of the K = 1280 level that carry a folded LayerNorm (rowstat) and / or GEGLU — qkv, q, GEGLU of
    // the third level and the middle block — on the wide tile's 256 x 256 form (gemm_xglds.hip EPI): 780-870 -> ~1000 TFLOP/s.  Taken
    // when the tile grid makes at least ~1.7 rounds of the chip well filled (the same criterion as the persistent kernel's >= 2 tiles
    // per CU, scaled to the twice-as-large tile); VMV_GEMM_XEPI=0 keeps the persistent kernel.
    if (gemm_policy() >= 2 && xglds_policy() && (geglu || p.rowstat) && p.ksplit <= 1 && !vmv_gemm_ln_inline(p) && total_steps >= 20 &&
        p.N % 256 == 0 && p.wgroup_rows == 0 && !p.out_fp32 && !p.rowvec && vmv_gemm_xglds_epi_ok(p, VMV_TILE_X256x256)) {
        static int xepi = -1;
        if (xepi < 0) { const char* e = getenv("VMV_GEMM_XEPI"); xepi = e ? atoi(e) : 1; }
        bool lin = true;
        for (int i = 0; i < p.nseg; ++i) lin = lin && p.seg[i].mode == VMV_SEG_LINEAR;
        const long tiles = (long)((p.M + 255) / 256) * (p.N / 256);
        const long rounds = (tiles + 255) / 256;
        if (xepi && lin && tiles >= 400 && (double)tiles / (double)(rounds * 256) >= 0.85) return VMV_TILE_X256x256;
    }
    {
        bool lin = p.ksplit <= 1 && total_steps <= 24 && p.M >= 16384;
        for (int i = 0; i < p.nseg; ++i) lin = lin && p.seg[i].mode == VMV_SEG_LINEAR;
        if (lin) {
            const int o = tile_override(geglu ? 0 : (p.N % 160 == 0 ? 1 : 2));
            if (o > 0) return o;
        }
    }
#if defined(VMV_EXPERIMENTS)
    if (gemm_policy() >= 2 && astat_policy() && p.N >= 640 && p.M >= 128 * 256 && vmv_gemm_astat_eligible(p))
        return (!geglu && p.N % 160 == 0) ? VMV_TILE_A128x160 : VMV_TILE_A128x128;
#endif
    auto padded = [&](int bn) { return ((p.N + bn - 1) / bn) * bn; };
    int best = VMV_TILE_128x128, best_pad = padded(128);
    if (!geglu && padded(160) <= best_pad) { best = VMV_TILE_128x160; best_pad = padded(160); }
    if (padded(64) < best_pad) { best = VMV_TILE_128x64; best_pad = padded(64); }
    if (p.M <= 64 && best == VMV_TILE_128x64) best = VMV_TILE_64x64;
    if (gemm_policy() >= 1 && p.ksplit > 1 && p.M > 64 && (best == VMV_TILE_128x128 || best == VMV_TILE_128x160)) {
        // split-K (small-M levels): the 4-wave LDS-DMA kernel instead of the register-staged one (same 128-row tiles) ...
        const int bn = best == VMV_TILE_128x128 ? 128 : 160;
        best = best == VMV_TILE_128x128 ? VMV_TILE_G128x128 : VMV_TILE_G128x160;
        // ... unless the 8-wave 256-row tiles times the split make one round of the chip (one block per CU): twice the FLOPs per
        // LDS-DMA byte.  M = 1920, N = 1280 (the UNet's fourth level): 64 tiles x 4 splits = 256 blocks — conv 749 -> 805,
        // conv 2560 -> 1280 893 -> 987, FF-down 559 -> 578 TFLOP/s (tools/experiments/run_l3_matrix.sh, same box)
        const long blocks256 = (long)((p.M + 255) / 256) * ((p.N + bn - 1) / bn) * p.ksplit;
        if (gemm_policy() >= 2 && p.M >= 1024 && blocks256 >= 192 && blocks256 <= 272)
            best = bn == 128 ? VMV_TILE_256x128 : VMV_TILE_256x160;
    }
#if defined(VMV_EXPERIMENTS)
    if (gemm_policy() >= 3 && p.ksplit <= 1 && !geglu && p.N % 160 == 0) {
        // wave-specialised persistent kernel (gemm_sglds.hip): 8 MFMA waves + 4 loader waves per CU.  Measured against every
        // other variant (tools/gemm_bench.py) it wins wherever its static round-robin over the 256 CUs is
        // balanced: +15-25 % on the L0 convs / temporal convs, +3-10 % on the L0 / L1 linears; it loses when the tile count
        // leaves the last round mostly empty (L2: 320 tiles = 1.25 rounds).
        const long items = (long)((p.M + 191) / 192) * (p.N / 160);
        const long rounds = (items + 255) / 256;
        bool any_gather = false;
        for (int i = 0; i < p.nseg; ++i) any_gather = any_gather || p.seg[i].mode != VMV_SEG_LINEAR;
        if (items >= 256 && (double)items / (double)(rounds * 256) >= 0.8 && (gemm_policy() == 3 || any_gather)) return VMV_TILE_S192x160;
    }
#endif
    if (gemm_policy() >= 1 && p.ksplit <= 1 && (best == VMV_TILE_128x128 || best == VMV_TILE_128x160)) {
        // the 256-row kernel runs one block per CU: use it when its grid still fills the 256 CUs well
        const int bn = best == VMV_TILE_128x128 ? 128 : 160;
        const long tiles = (long)((p.M + 255) / 256) * ((p.N + bn - 1) / bn);
        const long waves = (tiles + 255) / 256;
        const bool big = tiles >= 240 && (double)tiles / (double)(waves * 256) >= 0.8;
        // Short reductions over plain rows (the transformer blocks' linears, K = C .. 4C at the two large levels): the tile's
        // fill, epilogue round trips and store drain are as long as its main loop, so the persistent kernel — ring kept
        // full across tiles, epilogue loads prefetched under the last MFMAs — wins by 10-40 % once every CU gets >= 2 tiles
        // (measured per shape: DESIGN.md §7).  Long reductions and conv gathers stay on the one-tile-per-block kernel.
        bool linear = true;
        for (int i = 0; i < p.nseg; ++i) linear = linear && p.seg[i].mode == VMV_SEG_LINEAR;
        const bool p160 = best == VMV_TILE_128x160;
        const long ptiles = (long)((p.M + (p160 ? 191 : 255)) / (p160 ? 192 : 256)) * ((p.N + bn - 1) / bn);
        if (gemm_policy() >= 2 && linear && total_steps <= 24 && ptiles >= 2 * 256)
            best = p160 ? VMV_TILE_P256x160 : VMV_TILE_P256x128;
        else if (big)
            best = best == VMV_TILE_128x128 ? VMV_TILE_256x128 : VMV_TILE_256x160;
        else if (p.M > 64)
            best = best == VMV_TILE_128x128 ? VMV_TILE_G128x128 : VMV_TILE_G128x160;
    }
    return best;
}

// pick_tile + the constraints of the optional features (in-loop LayerNorm statistics, grouped weights, folded LayerNorm): the
// configuration vmv_gemm launches first, or a negative VMV_E* code for a forced tile that cannot serve the request
int final_tile(const VmvGemmParams& p, int total_steps) {
    int picked = pick_tile(p, total_steps);
#if !defined(VMV_EXPERIMENTS)
    if (picked == VMV_TILE_S256x128 || picked == VMV_TILE_S192x160 || picked == VMV_TILE_S256x160 || picked == VMV_TILE_A128x160 ||
        picked == VMV_TILE_A128x128 || picked == VMV_TILE_W256x256 || picked == VMV_TILE_Y256x128) return VMV_EINVAL;
#endif
    if (picked == VMV_TILE_HALO) return vmv_conv_halo_supported(p) ? picked : VMV_EINVAL;
    if (picked == VMV_TILE_TFR) return vmv_gemm_tfr_supported(p) ? picked : VMV_EINVAL;
    if (picked == VMV_TILE_W256x256) return vmv_gemm_wreg_supported(p) ? picked : VMV_EINVAL;
    if (picked == VMV_TILE_TQA) return vmv_gemm_tqa_supported(p) && (p.tile == VMV_TILE_AUTO || p.tile == VMV_TILE_TQA) ? picked : VMV_EINVAL;
    const bool rs_tile = picked == VMV_TILE_RS || picked == VMV_TILE_RS512 || picked == VMV_TILE_RS256;
    if (rs_tile) return vmv_gemm_rs_supported(p) ? picked : VMV_EINVAL;      // (handles rowstat / colsum / grouped weights itself)
    if (p.gn_table) return VMV_EINVAL;                                       // (a forced tile that cannot fold the GroupNorm)
    if (vmv_gemm_ln_inline(p) && p.tile == VMV_TILE_AUTO)
        picked = vmv_gemm_rs_supported(p) ? VMV_TILE_RS
                                          : (p.epilogue != VMV_EPI_GEGLU && p.N % 160 == 0) ? VMV_TILE_P256x160 : VMV_TILE_P256x128;
    if (picked == VMV_TILE_RS) return picked;
    if (p.wgroup_rows > 0) {       // served by the generic kernel and the 128-column LDS-DMA kernels (256- / 128-row tiles)
        const bool ok = picked == VMV_TILE_128x128 || picked == VMV_TILE_128x64 || picked == VMV_TILE_64x64 || picked == VMV_TILE_256x128 ||
                        picked == VMV_TILE_G128x128 || picked == VMV_TILE_P256x128;
        if (p.tile != VMV_TILE_AUTO) {
            if (!ok) return VMV_EINVAL;
        } else if (picked == VMV_TILE_P256x160 || picked == VMV_TILE_X256x320 || picked == VMV_TILE_X256x256 || picked == VMV_TILE_X512x128) picked = VMV_TILE_P256x128;
        else if (picked == VMV_TILE_256x160) picked = VMV_TILE_256x128;
        else if (!ok) picked = VMV_TILE_G128x128;
    }
    if (p.rowstat && !((picked == VMV_TILE_X256x256 || picked == VMV_TILE_Y256x128) && vmv_gemm_xglds_epi_ok(p, picked)) && picked != VMV_TILE_A128x160 && picked != VMV_TILE_A128x128 && picked != VMV_TILE_P256x128 && picked != VMV_TILE_P256x160 && picked != VMV_TILE_Q128x128 &&
        picked != VMV_TILE_Q96x160 && picked != VMV_TILE_128x128 && picked != VMV_TILE_128x160 && picked != VMV_TILE_128x64 &&
        picked != VMV_TILE_64x64)
        picked = (p.epilogue != VMV_EPI_GEGLU && p.N % 160 == 0) ? VMV_TILE_P256x160 : VMV_TILE_P256x128;
    return picked;
}

bool ln_inline_ok(const VmvGemmParams& p) {
    if (!p.W || !p.out || !p.colsum || p.nseg != 1 || p.seg[0].mode != VMV_SEG_LINEAR || p.seg[0].k != p.ktot) return false;
    if (p.ksplit > 1 || p.out_fp32 || p.rowvec || p.residual) return false;
    const bool rs_forced = p.tile == VMV_TILE_RS || p.tile == VMV_TILE_RS512 || p.tile == VMV_TILE_RS256;
    if (p.tile != VMV_TILE_AUTO && p.tile != VMV_TILE_P256x128 && p.tile != VMV_TILE_P256x160 && !rs_forced) return false;
    const int n_out = p.epilogue == VMV_EPI_GEGLU ? p.N / 2 : p.N;
    if ((p.ldo & 7) || (n_out & 7) || !vmv_aligned16(p.out)) return false;                  // the staged epilogue
    if (rs_forced) return vmv_gemm_rs_supported(p);
    return vmv_gemm_rs_supported(p) || vmv_gemm_pglds_supported(p);
}

}  // namespace

extern "C" int vmv_has_experiments(void) {
#if defined(VMV_EXPERIMENTS)
    return 1;
#else
    return 0;
#endif
}

extern "C" int vmv_gemm_ln_inline_ok(const VmvGemmParams* pp) { return pp && ln_inline_ok(*pp) ? 1 : 0; }

extern "C" int vmv_gemm_rs_ok(const VmvGemmParams* pp) {
    if (!pp || pp->tile != VMV_TILE_AUTO || pp->nseg <= 0 || pp->nseg > VMV_MAX_SEGS) return 0;
    return gemm_policy() >= 2 && vmv_gemm_rs_preferred(*pp) ? 1 : 0;
}

extern "C" int vmv_gemm_tfr_ok(const VmvGemmParams* pp) {
    if (!pp || pp->tile != VMV_TILE_AUTO || pp->nseg <= 0 || pp->nseg > VMV_MAX_SEGS) return 0;
    return gemm_policy() >= 2 && vmv_gemm_tfr_preferred(*pp) ? 1 : 0;
}

extern "C" int vmv_gemm_tqa_ok(const VmvGemmParams* pp) {
    if (!pp || pp->tile != VMV_TILE_AUTO || pp->nseg <= 0 || pp->nseg > VMV_MAX_SEGS) return 0;
    return gemm_policy() >= 2 && vmv_gemm_tqa_preferred(*pp) ? 1 : 0;
}

extern "C" int vmv_gemm_pick_tile(const VmvGemmParams* pp) {
    if (!pp || pp->nseg <= 0 || pp->nseg > VMV_MAX_SEGS) return VMV_EINVAL;
    int total_steps = 0;
    for (int s = 0; s < pp->nseg; ++s) total_steps += (pp->seg[s].k + BK - 1) / BK;
    return final_tile(*pp, total_steps);
}

thread_local int vmv_dry_run = 0;
// everything vmv_gemm(p, stream) does on the host — argument validation, the tile policy (or the forced p->tile), the chosen launcher's
// own eligibility checks — without touching the device: VMV_OK iff the same call would launch.  Used at RECORD time for forced tiles
// (a stale tuned-table entry is ignored there instead of failing on the first replay) and by the CPU tests.
extern "C" int vmv_gemm_validate(const VmvGemmParams* pp) {
    vmv_dry_run = 1;
    const int rc = vmv_gemm(pp, nullptr);
    vmv_dry_run = 0;
    return rc;
}

extern "C" int vmv_gemm(const VmvGemmParams* pp, void* stream) {
    if (!pp) return VMV_ENULL;
    const VmvGemmParams& p = *pp;
    if (!p.W || !p.out) return VMV_ENULL;
    if (p.M <= 0 || p.N <= 0 || (p.N & 3) || p.nseg <= 0 || p.nseg > VMV_MAX_SEGS) return VMV_EINVAL;
    if (!vmv_aligned16(p.W) || !vmv_aligned16(p.out) || (p.ktot & 7)) return VMV_EALIGN;
    if (p.bias && !vmv_aligned16(p.bias)) return VMV_EALIGN;
    if (p.ldo & 3) return VMV_EALIGN;
    int ksum = 0, total_steps = 0, maxld = p.ldo;
    for (int s = 0; s < p.nseg; ++s) {
        const VmvGemmSeg& sg = p.seg[s];
        if (!sg.src) return VMV_ENULL;
        if (!vmv_aligned16(sg.src) || (sg.ld & 7) || (sg.k & 7) || sg.k <= 0) return VMV_EALIGN;
        if (sg.mode == VMV_SEG_SPATIAL && (p.OH <= 0 || p.OW <= 0 || p.IH <= 0 || p.IW <= 0 || p.stride <= 0)) return VMV_EINVAL;
        if (sg.mode == VMV_SEG_TEMPORAL && (p.F <= 0 || p.P <= 0)) return VMV_EINVAL;
        if (sg.mode < 0 || sg.mode > 2) return VMV_EINVAL;
        ksum += sg.k;
        if (sg.ld > maxld) maxld = sg.ld;
        total_steps += (sg.k + BK - 1) / BK;
    }
    if (ksum != p.ktot) return VMV_EINVAL;
    if (p.epilogue == VMV_EPI_GEGLU && (p.N & 31)) return VMV_EINVAL;
    if (p.epilogue < VMV_EPI_NONE || p.epilogue > VMV_EPI_TATTN) return VMV_EINVAL;
    if (p.epilogue == VMV_EPI_TATTN && !vmv_gemm_tqa_supported(p)) return VMV_EINVAL;
    if (p.rowvec && (p.rowvec_div <= 0 || (p.rowvec_ld & 3) || !vmv_aligned16(p.rowvec))) return VMV_EINVAL;
    if (p.residual && ((p.ldr & 3) || (((uintptr_t)p.residual) & 7))) return VMV_EALIGN;
    if (p.ksplit > 1 && (!p.workspace || !vmv_aligned16(p.workspace))) return VMV_ENULL;
    if ((long)p.M * (long)maxld >= (1L << 31) || (long)p.N * (long)p.ktot >= (1L << 31)) return VMV_ERANGE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rc;
    if (p.rowstat) {       // LayerNorm-folded GEMM: implemented by the persistent kernel's epilogue and the generic one
        if (!p.colsum) return VMV_ENULL;
        if ((((uintptr_t)p.rowstat) & 7) || !vmv_aligned16(p.colsum)) return VMV_EALIGN;
        if (p.ksplit > 1) return VMV_EINVAL;
        for (int s = 0; s < p.nseg; ++s) if (p.seg[s].mode != VMV_SEG_LINEAR) return VMV_EINVAL;
    }
    if (p.wgroup_rows != 0) {      // grouped weights: a tile (<= 256 rows) never straddles two groups; the groups' matrices are
                                   // addressed with 32-bit byte offsets from W
        if (p.wgroup_rows < 0 || (p.wgroup_rows & 255) || p.wgroup_stride < 0 || (p.wgroup_stride & 7) || p.ksplit > 1) return VMV_EINVAL;
        const long groups = ((long)p.M + p.wgroup_rows - 1) / p.wgroup_rows;
        if (((groups - 1) * p.wgroup_stride + (long)p.N * p.ktot) * 2 >= (1L << 31) - 65536) return VMV_ERANGE;
    }
    if (p.gn_table && !vmv_gemm_rs_supported(p) && !vmv_gemm_tfr_supported(p)) return VMV_EINVAL;      // folded GroupNorm: gemm_rs / gemm_tfr only
    if (p.gn_silu && !(p.gn_table && vmv_gemm_tfr_supported(p))) return VMV_EINVAL;
    const bool ln_inline = vmv_gemm_ln_inline(p);
    if (ln_inline) {       // statistics in the main loop: the persistent one-block-per-CU kernel, staged 16-bit output
        if (!ln_inline_ok(p)) return VMV_EINVAL;
        if (!vmv_aligned16(p.colsum)) return VMV_EALIGN;
    }
    const int picked = final_tile(p, total_steps);
    if (picked < 0) return picked;
    switch (picked) {
        case VMV_TILE_128x128: rc = launch_cfg<4, 4>(p, total_steps, st); break;
        case VMV_TILE_128x160:
            if (p.epilogue == VMV_EPI_GEGLU) return VMV_EINVAL;
            rc = launch_cfg<4, 5>(p, total_steps, st); break;
        case VMV_TILE_128x64: rc = launch_cfg<4, 2>(p, total_steps, st); break;
        case VMV_TILE_64x64: rc = launch_cfg<2, 2>(p, total_steps, st); break;
        case VMV_TILE_256x128:
            rc = vmv_gemm_glds_launch(p, total_steps, VMV_TILE_256x128, st);
            if (rc == VMV_GLDS_UNSUPPORTED) rc = launch_cfg<4, 4>(p, total_steps, st);
            break;
        case VMV_TILE_256x160:
            rc = vmv_gemm_glds_launch(p, total_steps, VMV_TILE_256x160, st);
            if (rc == VMV_GLDS_UNSUPPORTED) rc = launch_cfg<4, 5>(p, total_steps, st);
            break;
        case VMV_TILE_S256x128:
            rc = vmv_gemm_sglds_launch(p, total_steps, VMV_TILE_S256x128, st);
            if (rc == VMV_GLDS_UNSUPPORTED) rc = vmv_gemm_glds_launch(p, total_steps, VMV_TILE_256x128, st);
            if (rc == VMV_GLDS_UNSUPPORTED) rc = launch_cfg<4, 4>(p, total_steps, st);
            break;
        case VMV_TILE_S192x160:
        case VMV_TILE_S256x160:
            rc = vmv_gemm_sglds_launch(p, total_steps, picked, st);
            if (rc == VMV_GLDS_UNSUPPORTED) rc = vmv_gemm_glds_launch(p, total_steps, VMV_TILE_256x160, st);
            if (rc == VMV_GLDS_UNSUPPORTED) rc = launch_cfg<4, 5>(p, total_steps, st);
            break;
        case VMV_TILE_X256x320:
        case VMV_TILE_X256x256:
        case VMV_TILE_X256x128:
        case VMV_TILE_X512x128:
        case VMV_TILE_Y256x128:
            rc = vmv_gemm_xglds_launch(p, total_steps, picked, st);
            if (rc == VMV_GLDS_UNSUPPORTED && (p.rowstat || p.epilogue == VMV_EPI_GEGLU)) {      // the fused epilogues' other home
                if (p.tile != VMV_TILE_AUTO) return VMV_EINVAL;
                rc = vmv_gemm_pglds_launch(p, total_steps, VMV_TILE_P256x128, st);
                if (rc == VMV_GLDS_UNSUPPORTED) rc = launch_cfg<4, 4>(p, total_steps, st);
                break;
            }
            if (rc == VMV_GLDS_UNSUPPORTED) {
                if (p.tile != VMV_TILE_AUTO) return VMV_EINVAL;
                rc = vmv_gemm_glds_launch(p, total_steps, p.N % 160 == 0 ? VMV_TILE_256x160 : VMV_TILE_256x128, st);
                if (rc == VMV_GLDS_UNSUPPORTED) rc = p.N % 160 == 0 ? launch_cfg<4, 5>(p, total_steps, st) : launch_cfg<4, 4>(p, total_steps, st);
            }
            break;
        case VMV_TILE_TFR:
            rc = vmv_gemm_tfr_launch(p, st);
            if (rc == VMV_GLDS_UNSUPPORTED) return VMV_EINVAL;
            break;
        case VMV_TILE_W256x256:
            rc = vmv_gemm_wreg_launch(p, st);
            if (rc == VMV_GLDS_UNSUPPORTED) return VMV_EINVAL;
            break;
        case VMV_TILE_TQA:
            rc = vmv_gemm_tqa_launch(p, st);
            if (rc == VMV_GLDS_UNSUPPORTED) return VMV_EINVAL;
            break;
        case VMV_TILE_HALO:
            rc = vmv_conv_halo_launch(p, st);
            if (rc == VMV_GLDS_UNSUPPORTED) return VMV_EINVAL;
            break;
        case VMV_TILE_RS:
        case VMV_TILE_RS512:
        case VMV_TILE_RS256:
            rc = vmv_gemm_rs_launch(p, picked, st);
            if (rc == VMV_GLDS_UNSUPPORTED) return VMV_EINVAL;      // (final_tile checked eligibility: a forced row tile that does not exist)
            break;
        case VMV_TILE_A128x160:
        case VMV_TILE_A128x128:
            rc = vmv_gemm_astat_launch(p, picked, st);
            if (rc == VMV_GLDS_UNSUPPORTED) return VMV_EINVAL;      // (only reachable with a forced tile id)
            break;
        case VMV_TILE_Q128x128:
            rc = vmv_gemm_pglds_launch(p, total_steps, VMV_TILE_Q128x128, st);
            if (rc == VMV_GLDS_UNSUPPORTED && !p.rowstat) rc = vmv_gemm_glds_launch(p, total_steps, VMV_TILE_256x128, st);
            if (rc == VMV_GLDS_UNSUPPORTED) rc = launch_cfg<4, 4>(p, total_steps, st);
            break;
        case VMV_TILE_Q96x160:
            rc = vmv_gemm_pglds_launch(p, total_steps, VMV_TILE_Q96x160, st);
            if (rc == VMV_GLDS_UNSUPPORTED && !p.rowstat) rc = vmv_gemm_glds_launch(p, total_steps, VMV_TILE_256x160, st);
            if (rc == VMV_GLDS_UNSUPPORTED) rc = launch_cfg<4, 5>(p, total_steps, st);
            break;
        case VMV_TILE_P256x128:
            rc = vmv_gemm_pglds_launch(p, total_steps, VMV_TILE_P256x128, st);
            if (rc == VMV_GLDS_UNSUPPORTED && ln_inline) return VMV_EINVAL;
            if (rc == VMV_GLDS_UNSUPPORTED && !p.rowstat) rc = vmv_gemm_glds_launch(p, total_steps, VMV_TILE_256x128, st);
            if (rc == VMV_GLDS_UNSUPPORTED) rc = launch_cfg<4, 4>(p, total_steps, st);
            break;
        case VMV_TILE_P256x160:
            rc = vmv_gemm_pglds_launch(p, total_steps, VMV_TILE_P256x160, st);
            if (rc == VMV_GLDS_UNSUPPORTED && ln_inline) return VMV_EINVAL;
            if (rc == VMV_GLDS_UNSUPPORTED && !p.rowstat) rc = vmv_gemm_glds_launch(p, total_steps, VMV_TILE_256x160, st);
            if (rc == VMV_GLDS_UNSUPPORTED) rc = launch_cfg<4, 5>(p, total_steps, st);
            break;
        case VMV_TILE_PP256x128:
            rc = vmv_gemm_glds_launch(p, total_steps, VMV_TILE_PP256x128, st);
            if (rc == VMV_GLDS_UNSUPPORTED) rc = launch_cfg<4, 4>(p, total_steps, st);
            break;
        case VMV_TILE_PP256x160:
            rc = vmv_gemm_glds_launch(p, total_steps, VMV_TILE_PP256x160, st);
            if (rc == VMV_GLDS_UNSUPPORTED) rc = launch_cfg<4, 5>(p, total_steps, st);
            break;
        case VMV_TILE_G128x128:
            rc = vmv_gemm_glds_launch(p, total_steps, VMV_TILE_G128x128, st);
            if (rc == VMV_GLDS_UNSUPPORTED) rc = launch_cfg<4, 4>(p, total_steps, st);
            break;
        case VMV_TILE_G128x160:
            rc = vmv_gemm_glds_launch(p, total_steps, VMV_TILE_G128x160, st);
            if (rc == VMV_GLDS_UNSUPPORTED) rc = launch_cfg<4, 5>(p, total_steps, st);
            break;
        default: return VMV_EINVAL;
    }
    if (rc != VMV_OK) return rc;
    if (p.ksplit > 1) {      // deterministic second pass: sum the fp32 slabs and run the epilogue
        const long items = (long)p.M * (p.N / 4);
        int blocks = (int)((items + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        VMV_LAUNCH(gemm_splitk_reduce, dim3(blocks), dim3(256), 0, st, p);
        rc = vmv_launch_status();
    }
    return rc;
}

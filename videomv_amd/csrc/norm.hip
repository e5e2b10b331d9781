// norm.hip — GroupNorm(32) statistics / apply(+SiLU) and LayerNorm over channels-last bf16 rows (gfx950).
// All three are HBM-bound streaming kernels: 16-byte vector loads/stores (8 bf16 per lane), fp32 math,
// wave-shuffle + LDS reductions; partial sums are written per chunk and reduced in a fixed order, or (long stat groups)
// added to 64-bit fixed-point integer accumulators — either way results are bitwise reproducible run to run.
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace {

// ------------------------------------------------------------------------------------------------ GN stats
// grid = (nchunk, nstat).  Block (256 threads) owns `chunk_rows` rows of one stat group and ALL channels.
// Thread -> fixed column slot (8 channels) so sums stay in registers; lanes that share a column are combined
// through an LDS [lane_rows][C] array in fixed order; then 32 threads produce the per-group (sum, sumsq).
// Statistics are taken of d = x - pilot, the pilot of (stat group, channel group) being the group's first element (first row of the
// stat group, first channel of the group): shift-invariant — E[d^2] - E[d]^2 has nothing to cancel when |mean| >> sigma (x = 100 +
// N(0, 1)) — at the price of 32 extra 2-byte loads per block.  Stat-group totals (VmvGroupNormParams.totals) are TWO-LIMB 64-bit
// fixed point per value: an integer limb (quantum 1, |sum| < 2^63: fp16's largest squares times a whole sample fit) and a fraction
// limb (quantum 2^-40), so a block's fp32 partial sum is represented exactly whether the activations are 1e-3 or 3e3 (round 2's
// single limb had a fixed 2^-12 quantum: a group of 1e-3-sized activations lost its variance).  Integer addition
// commutes: the totals are bitwise reproducible whatever the arrival order.  Record per (stat, group): GN_REC int64 =
// { sum_hi, sumsq_hi, sum_lo, sumsq_lo, pilot (fp32 bits), pad x3 }, kept GN_NREP times (see the atomics below).
constexpr int GN_REC = 8;
constexpr int GN_NREP = 8;                      // replicas of a record: chunk c adds into replica c % 8 (below)
constexpr float GN_LO = 1099511627776.0f;       // 2^40

VMV_DEV void gn_add2(unsigned long long* rec, int which, float v) {
    const float fl = floorf(v);
    atomicAdd(rec + which, (unsigned long long)(long long)fl);
    atomicAdd(rec + 2 + which, (unsigned long long)(long long)((v - fl) * GN_LO));
}
VMV_DEV float gn_pilot(const VmvGroupNormParams& p, long row, int c) {          // element (row, channel c) of the (two-source) input
    const bool first = c < p.C0;
    const uint16_t* b = reinterpret_cast<const uint16_t*>(first ? p.x : p.x1);
    return elem_to_f32(b[row * (long)(first ? p.ld : p.ld1) + (first ? c : c - p.C0)]);
}

__global__ __launch_bounds__(256) void gn_stats_kernel(const VmvGroupNormParams p, const int nchunk) {
    VMV_KERNEL_ENTER();
    extern __shared__ __attribute__((aligned(16))) float sh[];
    const int C = p.C0 + p.C1;
    const int CS = C >> 3;                       // 16-byte column slots per row
    const int TPR = CS < 256 ? CS : 256;         // threads per row
    const int RPP = 256 / TPR;                   // rows per pass
    const int tid = threadIdx.x;
    const int rl = tid / TPR;                    // row lane
    const int cl = tid - rl * TPR;               // column lane
    const bool active = rl < RPP;
    const int stat = blockIdx.y, chunk = blockIdx.x;
    const long row0 = (long)stat * p.rows_per_stat + (long)chunk * p.chunk_rows;
    long row_end = row0 + p.chunk_rows;
    const long stat_end = (long)(stat + 1) * p.rows_per_stat;
    if (row_end > stat_end) row_end = stat_end;
    float* lsum = sh;                            // [RPP][C]
    float* lsq = sh + RPP * C;
    float* pil = sh + 2 * RPP * C;               // [32] pilots of this stat group
    const uint16_t* x0 = reinterpret_cast<const uint16_t*>(p.x);
    const uint16_t* x1 = reinterpret_cast<const uint16_t*>(p.x1);
    const int cpg_ = C >> 5;
#if defined(VMV_GN_ABLATE_PILOT)
    if (tid < 32) pil[tid] = 0.f;
#else
    if (tid < 32) pil[tid] = gn_pilot(p, (long)stat * p.rows_per_stat, tid * cpg_);
#endif
    __syncthreads();
    for (int cs = cl; cs < CS; cs += TPR) {
        float s[8], q[8], pl[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; pl[e] = pil[(cs * 8 + e) / cpg_]; }
        if (active) {
            const int c = cs * 8;
            const bool first = c < p.C0;
            const uint16_t* base = first ? (x0 + c) : (x1 + (c - p.C0));
            const long ld = first ? p.ld : p.ld1;
            long r = row0 + rl;
            for (; r + 7L * RPP < row_end; r += 8L * RPP) {      // 8 independent 16-B loads in flight per lane
                u32x4_t v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const u32x4_t*>(base + (r + (long)k * RPP) * ld);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float f[8];
                    unpack8(v[k], f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float d = f[e] - pl[e]; s[e] += d; q[e] += d * d; }
                }
            }
            for (; r + 3L * RPP < row_end; r += 4L * RPP) {
                u32x4_t v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const u32x4_t*>(base + (r + (long)k * RPP) * ld);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float f[8];
                    unpack8(v[k], f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float d = f[e] - pl[e]; s[e] += d; q[e] += d * d; }
                }
            }
            for (; r < row_end; r += RPP) {
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(base + r * ld);
                float f[8];
                unpack8(v, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = f[e] - pl[e]; s[e] += d; q[e] += d * d; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { lsum[rl * C + cs * 8 + e] = s[e]; lsq[rl * C + cs * 8 + e] = q[e]; }
        }
    }
    __syncthreads();
    if (tid < 32) {
        const int cpg = C >> 5;
        float s = 0.f, q = 0.f;
        for (int r = 0; r < RPP; ++r)
            for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { s += lsum[r * C + c]; q += lsq[r * C + c]; }
        if (p.totals) {        // two-limb fixed-point 64-bit atomics: order-independent, hence deterministic (vmv.h)
            // Atomics on one 64-B record serialise at ~20 ns each wherever they are resolved (measured: 256 chunks x 4 adds on
            // one record = +20 us on a 17-us pass, linear in the chunk count: tools/experiments/gn_bench.py), so a record is
            // kept GN_NREP times and consecutive chunks take consecutive replicas; integer adds: the fold stays exact
            unsigned long long* rec = reinterpret_cast<unsigned long long*>(p.totals) +
                                      (((long)stat * 32 + tid) * GN_NREP + (chunk & (GN_NREP - 1))) * GN_REC;
            gn_add2(rec, 0, s);
            gn_add2(rec, 1, q);
            if (chunk == 0) rec[4] = (unsigned long long)__float_as_uint(pil[tid]);       // (replica 0 carries the pilot)
        } else {
            float* out = p.partial + (((long)stat * nchunk + chunk) * 32 + tid) * 2;
            out[0] = s; out[1] = q;
        }
    }
}

// ------------------------------------------------------------------------------------------------ GN apply
// grid = (nblk, nstat).  Every block first folds the partial sums of its stat group (fixed order) into
// mean / rstd for the 32 groups and expands them to per-channel scale/shift tables in LDS, then streams its
// rows: y = [silu](x * scale[c] + shift[c]).
// table != nullptr (vmv_groupnorm_table, grid = (1, nstat)): the block writes its scale / shift tables there and stops.
__global__ __launch_bounds__(256) void gn_apply_kernel(const VmvGroupNormParams p, const int nchunk,
                                                       const int apply_rows, const int nstat, float* const table) {
    VMV_KERNEL_ENTER();
    extern __shared__ __attribute__((aligned(16))) float sh[];
    const int C = p.C0 + p.C1;
    const int CS = C >> 3;
    const int cpg = C >> 5;
    const int tid = threadIdx.x;
    const int stat = blockIdx.y;
    float* scale = sh;        // [C]
    float* shift = sh + C;    // [C]
    float* s_mean = sh + 2 * C;   // [32]  (all LDS in the one dynamic region: keeps the base 16-B aligned)
    float* s_rstd = s_mean + 32;  // [32]
    {   // fold the per-chunk partial sums: 8 lanes per group, fixed order (lane-strided sums, then xor-shuffles)
        const int g = tid >> 3, sub = tid & 7;
        float s = 0.f, q = 0.f;
        // fold_ranks R > 1: partial = [R][nstat][nchunk][64] (all-gathered shards); every rank folds in the same order
        const int R = p.fold_ranks > 1 ? p.fold_ranks : 1;
        if (p.totals) {      // two-limb integer totals per (rank, stat, group), each relative to that rank's pilot: moved to rank 0's
                             // pilot and summed in fp64 (R shards of equal size: frame-parallel pixel shards)
            static_assert(GN_NREP == 8, "one replica per sub-lane");
            const double nr = (double)p.rows_per_stat * (double)cpg;
            double S = 0.0, Q = 0.0, P0 = 0.0;
            for (int r = 0; r < R; ++r) {
                const long long* rec = reinterpret_cast<const long long*>(p.totals) +
                                       ((((long)r * nstat + stat) * 32 + g) * GN_NREP + sub) * GN_REC;
                long long l0 = rec[0], l1 = rec[1], l2 = rec[2], l3 = rec[3];
                const long long pb = __shfl(rec[4], tid & 56, 64);         // replica 0's
#pragma unroll
                for (int o = 1; o < 8; o <<= 1) {       // the 8 replicas: exact integer sums
                    l0 += __shfl_xor(l0, o, 64); l1 += __shfl_xor(l1, o, 64);
                    l2 += __shfl_xor(l2, o, 64); l3 += __shfl_xor(l3, o, 64);
                }
                const double sr = (double)l0 + (double)l2 * (1.0 / (double)GN_LO);
                const double qr = (double)l1 + (double)l3 * (1.0 / (double)GN_LO);
                const double P = (double)__uint_as_float((uint32_t)pb);
                if (r == 0) P0 = P;
                const double d = P - P0;
                S += sr + nr * d;
                Q += qr + 2.0 * d * sr + nr * d * d;
            }
            if (sub == 0) {
                const double n = nr * (double)R;
                const double m = S / n;
                double var = Q / n - m * m;
                var = var < 0.0 ? 0.0 : var;
                s_mean[g] = (float)(P0 + m);
                s_rstd[g] = (float)(1.0 / sqrt(var + (double)p.eps));
            }
        } else {             // per-chunk fp32 partial sums of x - pilot (the pilot is re-read: the same element the statistics pass used)
            const float* pp = p.partial + ((long)stat * nchunk * 32 + g) * 2;
            for (int c = sub; c < nchunk; c += 8) { s += pp[(long)c * 64]; q += pp[(long)c * 64 + 1]; }
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
            if (sub == 0) {
                const float n = (float)p.rows_per_stat * (float)cpg;
                const float m = s / n;
                float var = q / n - m * m;
                var = var < 0.f ? 0.f : var;
                s_mean[g] = gn_pilot(p, (long)stat * p.rows_per_stat, g * cpg) + m;
                s_rstd[g] = rsqrtf(var + p.eps);
            }
        }
    }
    __syncthreads();
    if (p.totals_clear && blockIdx.x == 0 && blockIdx.y == 0)     // the NEXT norm's accumulators (never the ones read above)
        for (int i = tid; i < p.clear_count; i += 256) p.totals_clear[i] = 0;
    for (int c = tid; c < C; c += 256) {
        const int g = c / cpg;
        const float sc = s_rstd[g] * p.gamma[c];
        scale[c] = sc;
        shift[c] = p.beta[c] - s_mean[g] * sc;
    }
    __syncthreads();
    if (table) {
        for (int c = tid; c < C; c += 256) {
            table[((long)stat * 2 + 0) * C + c] = scale[c];
            table[((long)stat * 2 + 1) * C + c] = shift[c];
        }
        return;
    }
    const long row0 = (long)stat * p.rows_per_stat + (long)blockIdx.x * apply_rows;
    long row_end = row0 + apply_rows;
    const long stat_end = (long)(stat + 1) * p.rows_per_stat;
    if (row_end > stat_end) row_end = stat_end;
    // thread -> fixed 16-byte column slot (its 8 scale/shift pairs live in registers), rows strided: no integer
    // division and no LDS traffic in the streaming loop, 4 independent loads in flight per lane.
    const int TPR = CS < 256 ? CS : 256;
    const int RPP = 256 / TPR;
    const int rl = tid / TPR, cl = tid - rl * TPR;
    const uint16_t* x0 = reinterpret_cast<const uint16_t*>(p.x);
    const uint16_t* x1 = reinterpret_cast<const uint16_t*>(p.x1);
    uint16_t* y = reinterpret_cast<uint16_t*>(p.y);
    if (rl < RPP) {
        for (int cs = cl; cs < CS; cs += TPR) {
            const int c = cs * 8;
            float a[8], b[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { a[e] = scale[c + e]; b[e] = shift[c + e]; }
            const bool first = c < p.C0;
            const uint16_t* src = first ? (x0 + c) : (x1 + (c - p.C0));
            const long ld = first ? p.ld : p.ld1;
            uint16_t* dst = y + c;
            auto stream = [&](auto silu_tag) {
                constexpr bool SILU = decltype(silu_tag)::value;
                auto xform = [&](const u32x4_t& v) {
                    float f[8];
                    unpack8(v, f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        f[e] = f[e] * a[e] + b[e];
                        if constexpr (SILU) f[e] = silu_f(f[e]);
                    }
                    return pack8(f);
                };
                long r = row0 + rl;
                for (; r + 3L * RPP < row_end; r += 4L * RPP) {
                    u32x4_t v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const u32x4_t*>(src + (r + (long)k * RPP) * ld);
#pragma unroll
                    for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x4_t*>(dst + (r + (long)k * RPP) * p.ldy) = xform(v[k]);
                }
                for (; r < row_end; r += RPP)
                    *reinterpret_cast<u32x4_t*>(dst + r * p.ldy) = xform(*reinterpret_cast<const u32x4_t*>(src + r * ld));
            };
            if (p.silu) stream(std::true_type{}); else stream(std::false_type{});
        }
    }
}

// ------------------------------------------------------------------------------------------------ GN fused (small groups)
// grid = (C / CW, nstat).  A block owns ALL rows of one stat group for CW channels (whole groups), stages them in LDS as
// they stream in, and normalises from the stage: one launch and one read of x where the general path needs stats (+ fold)
// + apply — at the small levels those are ~5-9 us of launch latency each for < 10 us of work.  Statistics are two-pass
// (the data is on chip): mean first, then the squared deviations.  Fixed reduction order: bitwise reproducible.
#ifndef VMV_GNF_UNROLL
#define VMV_GNF_UNROLL 4      // (A/B, round 5: 8 / 16 loads in flight per lane measured 1.29 / 1.31 ms per step against 1.28: not the bound.
                              //  Round 6: the whole slab staged by LDS-DMA — EVERY load of the block in flight before the first wait, one
                              //  memory round trip instead of seven dependent ones — measured the same again: 16.9 vs 17.4 us per launch in
                              //  the kernel trace, 47.43 / 47.53 vs 47.54 / 47.35 ms per step (profiles/r6_gnf_dma_step_ab.log), so that form
                              //  was removed.  A block is ONE wave per SIMD walking four dependent LDS passes: VALU / LDS latency, not memory.)
#endif
__global__ __launch_bounds__(256) void gn_fused_kernel(const VmvGroupNormParams p, const int CW, const int stamp) {
    VMV_KERNEL_ENTER();
    // stamp != 0 (VMV_GNF_STAMP=1, experiments): block (0, 0) writes s_memtime at its phase edges into p.partial (>= 8 x 8 bytes)
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(p.partial);
    auto mark = [&](int i) { if (stamp && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) stamps[i] = __builtin_readcyclecounter(); };
    mark(0);
    extern __shared__ __attribute__((aligned(16))) float sh[];
    const int C = p.C0 + p.C1;
    const int cpg = C >> 5;
    const int G = CW / cpg;                       // groups of this block
    const int SW = CW >> 3;                       // 16-byte slots per staged row
    const int TPR = SW < 256 ? SW : 256;
    const int RPP = 256 / TPR;
    const int tid = threadIdx.x;
    const int rl = tid / TPR, cl = tid - rl * TPR;
    const bool active = rl < RPP;
    const int stat = blockIdx.y, c0 = blockIdx.x * CW;
    const int rows = p.rows_per_stat;
    const long row0 = (long)stat * rows;
    u32x4_t* stage = reinterpret_cast<u32x4_t*>(sh);                  // [rows][SW]
    float* red = sh + (size_t)rows * SW * 4;                          // [RPP][CW]
    float* scale = red + RPP * CW;                                    // [CW]
    float* shift = scale + CW;                                        // [CW]
    float* s_mean = shift + CW;                                       // [32]
    float* s_rstd = s_mean + 32;                                      // [32]
    const uint16_t* x0 = reinterpret_cast<const uint16_t*>(p.x);
    const uint16_t* x1 = reinterpret_cast<const uint16_t*>(p.x1);
    // ---- pass 1: stream in, stage, column sums
    for (int cs = cl; cs < SW; cs += TPR) {
        float s[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = 0.f;
        if (active) {
            const int c = c0 + cs * 8;
            const bool first = c < p.C0;
            const uint16_t* base = first ? (x0 + c) : (x1 + (c - p.C0));
            const long ld = first ? p.ld : p.ld1;
            int r = rl;
            // GNF_UNR loads in flight per lane (more were tried: profiles/r5_gnf_unroll_ab.log)
            constexpr int GNF_UNR = VMV_GNF_UNROLL;
            for (; r + (GNF_UNR - 1) * RPP < rows; r += GNF_UNR * RPP) {
                u32x4_t v[GNF_UNR];
#pragma unroll
                for (int k = 0; k < GNF_UNR; ++k) v[k] = *reinterpret_cast<const u32x4_t*>(base + (row0 + r + k * RPP) * ld);
#pragma unroll
                for (int k = 0; k < GNF_UNR; ++k) {
                    stage[(size_t)(r + k * RPP) * SW + cs] = v[k];
                    float f[8];
                    unpack8(v[k], f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) s[e] += f[e];
                }
            }
            for (; r < rows; r += RPP) {
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(base + (row0 + r) * ld);
                stage[(size_t)r * SW + cs] = v;
                float f[8];
                unpack8(v, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += f[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) red[rl * CW + cs * 8 + e] = s[e];
        }
    }
    __syncthreads();
    mark(1);
    const float n = (float)rows * (float)cpg;
    // (round 6: folding the RPP row lanes with 256 / CW threads per column instead of one measured no gain — the phase is its four
    //  barriers and the shuffle tree, 2 960 vs 2 672 ticks, profiles/r6_gnf_stamps_fold.log)
    // group totals in two parallel stages (fixed order): thread c < CW folds the RPP row lanes of column c, then one wave
    // per group sums the group's cpg columns with a shuffle tree
    auto group_totals = [&](float* dst, auto finish) {
        float cs = 0.f;
        if (tid < CW)
            for (int r = 0; r < RPP; ++r) cs += red[r * CW + tid];
        __syncthreads();
        if (tid < CW) red[tid] = cs;
        __syncthreads();
        const int wv = tid >> 6, ln = tid & 63;
        for (int g = wv; g < G; g += 4) {
            float s = 0.f;
            for (int c = ln; c < cpg; c += 64) s += red[g * cpg + c];
            s = wave_sum(s);
            if (ln == 0) dst[g] = finish(s);
        }
        __syncthreads();
    };
    group_totals(s_mean, [&](float s) { return s / n; });
    mark(2);
    // ---- pass 2 (from the stage): squared deviations from the group mean
    for (int cs = cl; cs < SW; cs += TPR) {
        if (active) {
            float mu[8], q[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { mu[e] = s_mean[(cs * 8 + e) / cpg]; q[e] = 0.f; }
            for (int r = rl; r < rows; r += RPP) {
                float f[8];
                unpack8(stage[(size_t)r * SW + cs], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = f[e] - mu[e]; q[e] += d * d; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) red[rl * CW + cs * 8 + e] = q[e];
        }
    }
    __syncthreads();
    mark(3);
    group_totals(s_rstd, [&](float q) { return rsqrtf(q / n + p.eps); });
    mark(4);
    for (int c = tid; c < CW; c += 256) {
        const int g = c / cpg;
        const float sc = s_rstd[g] * p.gamma[c0 + c];
        scale[c] = sc;
        shift[c] = p.beta[c0 + c] - s_mean[g] * sc;
    }
    __syncthreads();
    mark(5);
    // ---- apply from the stage
    uint16_t* y = reinterpret_cast<uint16_t*>(p.y);
    if (active) {
        for (int cs = cl; cs < SW; cs += TPR) {
            float a[8], b[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { a[e] = scale[cs * 8 + e]; b[e] = shift[cs * 8 + e]; }
            uint16_t* dst = y + c0 + cs * 8;
            auto run = [&](auto silu_tag) {
                constexpr bool SILU = decltype(silu_tag)::value;
                for (int r = rl; r < rows; r += RPP) {
                    float f[8];
                    unpack8(stage[(size_t)r * SW + cs], f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        f[e] = f[e] * a[e] + b[e];
                        if constexpr (SILU) f[e] = silu_f(f[e]);
                    }
                    *reinterpret_cast<u32x4_t*>(dst + (row0 + r) * p.ldy) = pack8(f);
                }
            };
            if (p.silu) run(std::true_type{}); else run(std::false_type{});
        }
    }
    mark(6);
}

// ------------------------------------------------------------------------------------------------ LayerNorm
// LPR lanes per row (64 / LPR rows per wave, 4 waves per block): a 320-channel row is only 40 16-byte slots, so with a
// whole wave per row 24 lanes idle and too few bytes are in flight; LPR = 8 / 16 / 32 / 64 for C = 320 / 640 / 1280 /
// 2048 keeps every lane on <= 5 slots.  The row lives in registers, two-pass mean / variance, shuffles over the LPR lanes.
constexpr int LN_MAX_IT = 5;   // C <= 5 * 64 * 8 = 2560
template <int LPR>
__global__ __launch_bounds__(256) void layernorm_kernel(const VmvLayerNormParams p) {
    VMV_KERNEL_ENTER();
    constexpr int RPW = 64 / LPR;                        // rows per wave
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR;
    const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const bool live = row < p.rows;
    const int CS = p.C >> 3;
    const uint16_t* x = reinterpret_cast<const uint16_t*>(p.x) + (live ? row : 0) * p.ldx;
    float f[LN_MAX_IT][8];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < LN_MAX_IT; ++it) {
        const int cs = sub + it * LPR;
        if (cs < CS) {
            unpack8(*reinterpret_cast<const u32x4_t*>(x + cs * 8), f[it]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += f[it][e];
        }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)p.C;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < LN_MAX_IT; ++it) {
        const int cs = sub + it * LPR;
        if (cs < CS) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = f[it][e] - mean; q += d * d; }
        }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = rsqrtf(q / (float)p.C + p.eps);
    if (!live) return;
    if (p.stats_out) {                                   // statistics pass of a LayerNorm folded into its consumer GEMM
        if (sub == 0) *reinterpret_cast<float2*>(p.stats_out + row * 2) = make_float2(mean, rstd);
        return;
    }
    uint16_t* y = reinterpret_cast<uint16_t*>(p.y) + row * p.ldy;
#pragma unroll
    for (int it = 0; it < LN_MAX_IT; ++it) {
        const int cs = sub + it * LPR;
        if (cs < CS) {
            const int c = cs * 8;
            const f32x4_t g0 = *reinterpret_cast<const f32x4_t*>(p.gamma + c);
            const f32x4_t g1 = *reinterpret_cast<const f32x4_t*>(p.gamma + c + 4);
            const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(p.beta + c);
            const f32x4_t b1 = *reinterpret_cast<const f32x4_t*>(p.beta + c + 4);
            float o[8];
            o[0] = (f[it][0] - mean) * rstd * g0.x + b0.x; o[1] = (f[it][1] - mean) * rstd * g0.y + b0.y;
            o[2] = (f[it][2] - mean) * rstd * g0.z + b0.z; o[3] = (f[it][3] - mean) * rstd * g0.w + b0.w;
            o[4] = (f[it][4] - mean) * rstd * g1.x + b1.x; o[5] = (f[it][5] - mean) * rstd * g1.y + b1.y;
            o[6] = (f[it][6] - mean) * rstd * g1.z + b1.z; o[7] = (f[it][7] - mean) * rstd * g1.w + b1.w;
            *reinterpret_cast<u32x4_t*>(y + c) = pack8(o);
        }
    }
}

// ------------------------------------------------------------------------------------------------ row softmax
// One wave per row (4 rows per block); three passes over an L2-resident fp32 row (max, sum of exp2, write bf16).
__global__ __launch_bounds__(256) void softmax_rows_kernel(const VmvSoftmaxParams p) {
    VMV_KERNEL_ENTER();
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const float* s = p.s + row * p.lds;
    const float sc = p.scale * 1.44269504088896341f;
    const int n4 = p.n >> 2;
    float mx = -3.0e38f;
    for (int i = lane; i < n4; i += 64) {
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(s + 4 * i);
        mx = fmaxf(fmaxf(fmaxf(mx, v.x), fmaxf(v.y, v.z)), v.w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    // scale may be negative in principle; VideoMV's is positive, so max(scale*s) = scale*max(s)
    const float m2 = mx * sc;
    float sum = 0.f;
    for (int i = lane; i < n4; i += 64) {
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(s + 4 * i);
        sum += __builtin_amdgcn_exp2f(v.x * sc - m2) + __builtin_amdgcn_exp2f(v.y * sc - m2) +
               __builtin_amdgcn_exp2f(v.z * sc - m2) + __builtin_amdgcn_exp2f(v.w * sc - m2);
    }
    const float inv = 1.0f / wave_sum(sum);
    uint16_t* out = reinterpret_cast<uint16_t*>(p.p) + row * p.ldp;
    for (int i = lane; i < n4; i += 64) {
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(s + 4 * i);
        u32x2_t w;
        w.x = pack_elem2(__builtin_amdgcn_exp2f(v.x * sc - m2) * inv, __builtin_amdgcn_exp2f(v.y * sc - m2) * inv);
        w.y = pack_elem2(__builtin_amdgcn_exp2f(v.z * sc - m2) * inv, __builtin_amdgcn_exp2f(v.w * sc - m2) * inv);
        *reinterpret_cast<u32x2_t*>(out + 4 * i) = w;
    }
}

int gn_check(const VmvGroupNormParams& p) {
    if (!p.x || (!p.partial && !p.totals)) return VMV_ENULL;
    const int C = p.C0 + p.C1;
    if (C <= 0 || (C & 31) || (p.C0 & 7) || (p.C1 & 7)) return VMV_EINVAL;
    if (p.C1 > 0 && !p.x1) return VMV_ENULL;
    if (p.rows <= 0 || p.rows_per_stat <= 0 || (p.rows % p.rows_per_stat) || p.chunk_rows <= 0) return VMV_EINVAL;
    if (!vmv_aligned16(p.x) || (p.ld & 7) || (p.C1 > 0 && (!vmv_aligned16(p.x1) || (p.ld1 & 7)))) return VMV_EALIGN;
    if (C > 4096) return VMV_ERANGE;
    return VMV_OK;
}

}  // namespace

extern "C" int vmv_groupnorm_stats(const VmvGroupNormParams* pp, void* stream) {
    if (!pp) return VMV_ENULL;
    const VmvGroupNormParams& p = *pp;
    int rc = gn_check(p);
    if (rc != VMV_OK) return rc;
    const int C = p.C0 + p.C1;
    const int CS = C >> 3;
    const int TPR = CS < 256 ? CS : 256;
    const int RPP = 256 / TPR;
    const int nstat = p.rows / p.rows_per_stat;
    const int nchunk = (p.rows_per_stat + p.chunk_rows - 1) / p.chunk_rows;
    const size_t shbytes = ((size_t)2 * RPP * C + 32) * sizeof(float);
    if (p.totals && (((uintptr_t)p.totals) & 7)) return VMV_EALIGN;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunk, nstat), dim3(256), shbytes, reinterpret_cast<hipStream_t>(stream), p, nchunk);
    return vmv_launch_status();
}

extern "C" int vmv_groupnorm_apply(const VmvGroupNormParams* pp, void* stream) {
    if (!pp) return VMV_ENULL;
    const VmvGroupNormParams& p = *pp;
    int rc = gn_check(p);
    if (rc != VMV_OK) return rc;
    if (!p.y || !p.gamma || !p.beta) return VMV_ENULL;
    if (!vmv_aligned16(p.y) || (p.ldy & 7) || !vmv_aligned16(p.gamma) || !vmv_aligned16(p.beta)) return VMV_EALIGN;
    if (p.fold_ranks > 1 && !p.totals) return VMV_EINVAL;      // shards are folded through the totals records (they carry the pilots)
    const int C = p.C0 + p.C1;
    const int nstat = p.rows / p.rows_per_stat;
    const int nchunk = (p.rows_per_stat + p.chunk_rows - 1) / p.chunk_rows;
    // ~64 KB of input per block: >= 1k blocks in flight at the large levels, table set-up amortised
    int apply_rows = (65536 / (C * 2));
    if (apply_rows < 1) apply_rows = 1;
    const int nblk = (p.rows_per_stat + apply_rows - 1) / apply_rows;
    hipLaunchKernelGGL(gn_apply_kernel, dim3(nblk, nstat), dim3(256), (size_t)(2 * C + 64) * sizeof(float),
                       reinterpret_cast<hipStream_t>(stream), p, nchunk, apply_rows, nstat, nullptr);
    return vmv_launch_status();
}

extern "C" int vmv_groupnorm_table(const VmvGroupNormParams* pp, void* stream) {
    if (!pp) return VMV_ENULL;
    const VmvGroupNormParams& p = *pp;
    int rc = gn_check(p);
    if (rc != VMV_OK) return rc;
    if (!p.y || !p.gamma || !p.beta) return VMV_ENULL;
    if (p.silu) return VMV_EINVAL;                         // an activation cannot ride in an affine table
    if ((((uintptr_t)p.y) & 15) || !vmv_aligned16(p.gamma) || !vmv_aligned16(p.beta)) return VMV_EALIGN;
    if (p.fold_ranks > 1 && !p.totals) return VMV_EINVAL;
    const int C = p.C0 + p.C1;
    const int nstat = p.rows / p.rows_per_stat;
    const int nchunk = (p.rows_per_stat + p.chunk_rows - 1) / p.chunk_rows;
    hipLaunchKernelGGL(gn_apply_kernel, dim3(1, nstat), dim3(256), (size_t)(2 * C + 64) * sizeof(float),
                       reinterpret_cast<hipStream_t>(stream), p, nchunk, 1, nstat, reinterpret_cast<float*>(p.y));
    return vmv_launch_status();
}

extern "C" int vmv_groupnorm_fused(const VmvGroupNormParams* pp, int32_t cols, void* stream) {
    if (!pp) return VMV_ENULL;
    VmvGroupNormParams p = *pp;
    if (!p.partial) p.partial = reinterpret_cast<float*>(const_cast<void*>(p.x));      // (unused here; gn_check wants it non-NULL)
    if (p.chunk_rows <= 0) p.chunk_rows = 1;
    int rc = gn_check(p);
    if (rc != VMV_OK) return rc;
    if (!p.y || !p.gamma || !p.beta) return VMV_ENULL;
    if (!vmv_aligned16(p.y) || (p.ldy & 7) || !vmv_aligned16(p.gamma) || !vmv_aligned16(p.beta)) return VMV_EALIGN;
    const int C = p.C0 + p.C1, cpg = C >> 5;
    if (cols <= 0 || (cols & 7) || (cols % cpg) || (C % cols)) return VMV_EINVAL;
    if (cols > 256) return VMV_ERANGE;              // the column fold is one thread per column of a 256-thread block
    if (p.fold_ranks > 1) return VMV_EINVAL;
    if ((long)p.rows_per_stat * cols * 2 > VMV_GN_FUSED_BYTES) return VMV_ERANGE;
    const int SW = cols >> 3;
    const int TPR = SW < 256 ? SW : 256, RPP = 256 / TPR;
    const size_t shbytes = (size_t)p.rows_per_stat * cols * 2 + (size_t)(RPP * cols + 2 * cols + 64) * sizeof(float);
    static std::atomic<unsigned long long> attr{0};
    if (const int rc_attr = vmv_lds_attr_once(attr, reinterpret_cast<const void*>(&gn_fused_kernel), 160 * 1024)) return rc_attr;
    if (shbytes > 160 * 1024) return VMV_ERANGE;
    static int stamp_env = -1;
    if (stamp_env < 0) { const char* e = getenv("VMV_GNF_STAMP"); stamp_env = e ? atoi(e) : 0; }
    hipLaunchKernelGGL(gn_fused_kernel, dim3(C / cols, p.rows / p.rows_per_stat), dim3(256), shbytes,
                       reinterpret_cast<hipStream_t>(stream), p, cols, stamp_env && pp->partial ? 1 : 0);
    return vmv_launch_status();
}

extern "C" int vmv_layernorm(const VmvLayerNormParams* pp, void* stream) {
    if (!pp) return VMV_ENULL;
    const VmvLayerNormParams& p = *pp;
    if (!p.x) return VMV_ENULL;
    if (!p.stats_out && (!p.y || !p.gamma || !p.beta)) return VMV_ENULL;
    if (p.rows <= 0 || p.C <= 0 || (p.C & 7)) return VMV_EINVAL;
    if (p.C > LN_MAX_IT * 64 * 8) return VMV_ERANGE;
    if (!vmv_aligned16(p.x) || (p.ldx & 7)) return VMV_EALIGN;
    if (p.stats_out) { if (((uintptr_t)p.stats_out) & 7) return VMV_EALIGN; }
    else if (!vmv_aligned16(p.y) || (p.ldy & 7) || !vmv_aligned16(p.gamma) || !vmv_aligned16(p.beta)) return VMV_EALIGN;
    const int CS = p.C >> 3;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    auto go = [&](auto lpr_tag) {
        constexpr int LPR = decltype(lpr_tag)::value;
        const int rows_per_block = 4 * (64 / LPR);
        const int blocks = (p.rows + rows_per_block - 1) / rows_per_block;
        hipLaunchKernelGGL(layernorm_kernel<LPR>, dim3(blocks), dim3(256), 0, st, p);
    };
    if (CS <= 8 * LN_MAX_IT) go(std::integral_constant<int, 8>{});
    else if (CS <= 16 * LN_MAX_IT) go(std::integral_constant<int, 16>{});
    else if (CS <= 32 * LN_MAX_IT) go(std::integral_constant<int, 32>{});
    else go(std::integral_constant<int, 64>{});
    return vmv_launch_status();
}

extern "C" int vmv_softmax_rows(const VmvSoftmaxParams* pp, void* stream) {
    if (!pp) return VMV_ENULL;
    const VmvSoftmaxParams& p = *pp;
    if (!p.s || !p.p) return VMV_ENULL;
    if (p.rows <= 0 || p.n <= 0 || (p.n & 3) || p.scale <= 0.f) return VMV_EINVAL;
    if (!vmv_aligned16(p.s) || (p.lds & 3) || (((uintptr_t)p.p) & 7) || (p.ldp & 3)) return VMV_EALIGN;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((p.rows + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
    return vmv_launch_status();
}

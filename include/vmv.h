/* vmv.h — C ABI of libvmv_hip.so: the MI355X (gfx950) kernels behind VideoMV's DDIM denoising hot path.
 *
 * The reference (alibaba/VideoMV) has NO native code of its own: its hot path bottoms out in PyTorch ops that
 * dispatch to cuDNN / cuBLAS / xformers.  Each entry point below therefore replaces a
 * family of those leaf calls; the reference call sites are cited per function (paths relative to the
 * reference root).  The Python host (videomv_amd/) binds these with ctypes — see INTEGRATION.md.
 *
 * Contract for every launcher:
 *   - plain pointers + sizes only; all pointers are caller-owned DEVICE memory; nothing is allocated or freed;
 *   - no synchronisation, no global state: work is enqueued on `stream` (a hipStream_t passed as void*),
 *     so calls are hipGraph-capturable and thread-safe across streams;
 *   - returns 0 on success, a negative VMV_E* code for argument violations (reported, never asserted), or the
 *     positive hipError_t of a failed launch.
 * Activations are channels-last "rows": a [rows, C] elem (fp16 | bf16, vmv_elem_type()) matrix whose row index enumerates (batch, frame, y, x)
 * (see DESIGN.md §3); weights are elem [N][K] (output-channel major, reduction contiguous).
 */
#ifndef VMV_H
#define VMV_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VMV_OK            0
#define VMV_EINVAL       -1   /* bad dimension / flag combination */
#define VMV_EALIGN       -2   /* pointer or leading dimension not 16-byte aligned */
#define VMV_ENULL        -3   /* required pointer is NULL */
#define VMV_ERANGE       -4   /* size outside what the kernel supports */
#define VMV_ECOMM        -5   /* an RCCL call of vmv_comm_* failed */

#define VMV_ABI_VERSION   11
int vmv_abi_version(void);
/* The 16-bit storage / MFMA operand type ("elem") this build of the library computes in.  The same sources are
 * compiled once per type: libvmv_hip_f16.so (VMV_ELEM_F16: IEEE fp16 — the default; the reference's own half mode,
 * `use_fp16` + autocast, configs/i2vgen_xl_infer.yaml) and libvmv_hip_bf16.so (VMV_ELEM_BF16).  Every "elem" pointer
 * below is an array of that type; accumulation, statistics and softmax are fp32 in both builds. */
#define VMV_ELEM_F16      0
#define VMV_ELEM_BF16     1
int vmv_elem_type(void);
/* 1 if this build carries the experiment kernels (make EXPERIMENTS=1: VMV_TILE_S*, VMV_TILE_A*, the ablation hooks), else 0:
 * vmv_gemm then answers VMV_EINVAL to those forced tile ids */
int vmv_has_experiments(void);
/* sizeof() of the argument blocks, so a foreign-language binding can verify its struct layout:
 * which = VMV_OP_* (GN_STATS/GN_APPLY share a block), 100 = VmvDdimParams, 101 = VmvGemmSeg, 102 = VmvSeqMap,
 * 103 = VmvGsParams, 104 = VmvGsBatchParams */
int vmv_sizeof(int which);
/* human-readable text for a code returned by any launcher (VMV_E* or hipError_t) */
const char* vmv_error_string(int code);

/* ------------------------------------------------------------------------------------------------------
 * Implicit-GEMM: out[M,N] = epilogue( sum_s  gather_s(A_s)[M,K_s] * W[:, koff_s : koff_s+K_s]^T )
 * One kernel covers (reference call sites):
 *   nn.Linear / Conv1d(k=1) / 1x1 conv ............ util.py:223-227,337,351,546,573,688,1016,1032
 *   Conv2d 3x3 s1 p1, s2 p1, nearest-x2 + 3x3 ..... util.py:651,677,749,595-606; unet_t2v.py:169,264
 *   Conv3d (3,1,1) zero-padded over frames ......... util.py:1360-1375
 *   VAE decoder convs .............................. autoencoder.py:582-687
 * A "segment" is one (source tensor, tap) pair; the K loop walks the segments in order, so im2col, the
 * channel-concat of decoder ResBlocks (torch.cat at unet_t2v.py:361) and the fused 1x1 skip connection
 * (util.py:720) never materialise.
 * ---------------------------------------------------------------------------------------------------- */
#define VMV_MAX_SEGS 24
#define VMV_SEG_LINEAR   0   /* source row = m */
#define VMV_SEG_SPATIAL  1   /* m -> (n, oy, ox); source pixel (oy*stride+d0, ox*stride+d1) in the (up-sampled) image */
#define VMV_SEG_TEMPORAL 2   /* m -> (b, f, p); source row m + d0*P if 0 <= f+d0 < F else zero */

typedef struct {
    const void* src;     /* elem rows; row stride `ld` elements                                  */
    int32_t ld;          /* elements; multiple of 8                                              */
    int32_t k;           /* channels taken from each source row (multiple of 8)                   */
    int32_t mode;        /* VMV_SEG_*                                                            */
    int32_t d0, d1;      /* tap offsets (dy,dx) or (dt,-)                                        */
    int32_t _pad;
} VmvGemmSeg;

#define VMV_EPI_NONE   0
#define VMV_EPI_GEGLU  1   /* N counts x|gate pairs interleaved in 16-column blocks; writes N/2 columns: x*gelu_erf(gate) */
#define VMV_EPI_TATTN  2   /* q | k | v projection + per-pixel TEMPORAL attention in one launch (gemm_tqa.hip; TemporalTransformer, util.py:1043-1089,
                              230-268): rows are (sample, frame, pixel) with geometry F, P; W = [heads][q | k | v][64][ktot] (head-major,
                              N = 192 * heads); out[m][64 h + d] = sum_f' softmax_f'(q_h(m) . k_h(m') * epi_scale) v_h(m')[d] over the F
                              frames m' of row m's pixel; q, k, v never reach memory.  Ask vmv_gemm_tqa_ok() first.                       */
#define VMV_ACT_NONE   0
#define VMV_ACT_SILU   1
#define VMV_ACT_GELU   2   /* exact (erf) GELU: nn.GELU() of the CLIP text tower's MLP */

typedef struct {
    int32_t M, N;            /* output rows / weight rows (N multiple of 4)                         */
    int32_t nseg;
    int32_t ktot;            /* = sum of seg.k = row stride of W                                   */
    VmvGemmSeg seg[VMV_MAX_SEGS];
    const void* W;           /* elem [N][ktot]                                                     */
    const float* bias;       /* [N] or NULL                                                        */
    const float* rowvec;     /* optional [M / rowvec_div][rowvec_ld] fp32 added per row group (time embedding) */
    int32_t rowvec_div, rowvec_ld;
    const void* residual;    /* optional elem [M][ldr] added last                                  */
    int32_t ldr;
    int32_t epilogue;        /* VMV_EPI_*                                                          */
    int32_t act;             /* VMV_ACT_* applied after bias/rowvec, before residual               */
    int32_t out_fp32;        /* 0: elem output, 1: fp32 output                                     */
    void* out;               /* [M][ldo]                                                           */
    int32_t ldo;
    /* geometry for SPATIAL segments */
    int32_t OH, OW, IH, IW, stride, ups;
    /* geometry for TEMPORAL segments */
    int32_t F, P;
    /* split-K: ksplit > 1 needs workspace of ksplit*M*N floats; 0/1 = off */
    int32_t ksplit;
    float* workspace;
    int32_t tile;            /* 0 = auto; else VMV_TILE_* to force a configuration                 */
    float res_scale;         /* the residual is added as res_scale * residual; 0 means 1 (LGM's (x + res) * sqrt(.5),
                                core/unet.py:99,49, with the weights pre-scaled on the host)                    */
    /* LayerNorm folded into the GEMM (BasicTransformerBlock: norm -> Linear, util.py:520-546).  For y = W LN(x) + b with
     * LN(x) = (x - mean) * rstd * gamma + beta the host packs W' = W diag(gamma), bias' = b + W beta and
     * colsum[n] = sum_k W'[n][k] (of the elem-rounded W'), the GEMM runs on the RAW rows, and the epilogue computes
     *     rstd[m] * (acc[m][n] - mean[m] * colsum[n]) + bias'[n]
     * with rowstat = fp32 [M][2] (mean, rstd) from vmv_layernorm(stats_out).  Linear segments only, no split-K.       */
    const float* rowstat;
    const float* colsum;     /* fp32 [N] (pre-GEGLU numbering), required with rowstat */
    /* rowstat == NULL, colsum != NULL, ln_eps > 0: the same folded LayerNorm with (mean, rstd) of every row accumulated in
     * the GEMM's own main loop from the A fragments it multiplies (no statistics pass, no rowstat traffic); the K range
     * must be the normalised row: ONE linear segment, k == ktot == LayerNorm width, no split-K, no residual / rowvec, 16-bit output.  Served by the
     * row-stationary kernel (K = 320 / 640: statistics from the resident rows, two-pass) and the persistent kernel: ask
     * vmv_gemm_ln_inline_ok() first, vmv_gemm returns VMV_EINVAL otherwise.                                           */
    float ln_eps;
    /* Grouped weights (batched small GEMMs in ONE launch: the VAE's single-head attention, autoencoder.py:366-390, runs
     * Q K^T and P V of all frames at once): output rows [g * wgroup_rows, (g + 1) * wgroup_rows) multiply the weight matrix at
     * W + g * wgroup_stride elements (same [N][ktot] layout for every group).  wgroup_rows = 0: one W for all rows.
     * wgroup_rows must be a multiple of 256 (a tile never straddles two groups); no split-K; served by the 128-column
     * LDS-DMA kernels (vmv_gemm returns VMV_EINVAL for other forced tiles).                                          */
    int32_t wgroup_rows;
    int64_t wgroup_stride;
    /* GroupNorm folded into the GEMM's A rows (SpatialTransformer / TemporalTransformer: norm -> proj_in, util.py:354-360, 1043-1050):
     * gn_table = fp32 [nstat][2][ktot] — per stat group (rows [s * gn_rows_per_stat, (s + 1) * gn_rows_per_stat)) the per-channel
     * scale then shift of the norm, written by vmv_groupnorm_table — and the GEMM multiplies elem(x * scale + shift), the very
     * values vmv_groupnorm_apply would have stored: the normalised tensor is never written or re-read.  Row-stationary kernel
     * only (ask vmv_gemm_rs_ok; K = 320 / 640, one linear segment, no folded LayerNorm), gn_rows_per_stat a multiple of 16 and
     * >= 512; vmv_gemm returns VMV_EINVAL otherwise.  NULL = off.                                                       */
    const float* gn_table;
    int32_t gn_rows_per_stat;
    /* Frame-resident temporal convolution (VMV_TILE_TFR; TemporalConvBlock_v2, util.py:1357-1392: GroupNorm over all frames -> SiLU ->
     * Conv3d (3,1,1)): with three TEMPORAL segments (dt = -1, 0, +1 of ONE source) gn_table folds that norm into the convolution's A
     * path — gn_table = [samples][2][C] with C = the segment width (not ktot), gn_rows_per_stat = F * P (one stat group per sample) —
     * and gn_silu != 0 applies SiLU after the affine: the GEMM multiplies elem(silu(x * scale + shift)), the values
     * vmv_groupnorm_apply(silu = 1) would have stored, with the zero padding of frames -1 / F applied AFTER the norm (as Conv3d pads
     * the normalised tensor).  Ask vmv_gemm_tfr_ok() first.  gn_silu is ignored by the row-stationary kernel (must be 0 there).    */
    int32_t gn_silu;
    /* VMV_EPI_TATTN: the softmax scale (1 / sqrt(head_dim) = 0.125); ignored by every other epilogue.  (ABI 10) */
    float epi_scale;
} VmvGemmParams;

#define VMV_TILE_AUTO     0
#define VMV_TILE_128x128  1
#define VMV_TILE_128x160  2
#define VMV_TILE_128x64   3
#define VMV_TILE_64x64    4
#define VMV_TILE_256x128  5   /* 8-wave LDS-DMA ring kernel (gemm_glds.hip) */
#define VMV_TILE_256x160  6
#define VMV_TILE_G128x128 7   /* 4-wave LDS-DMA kernel, 2-stage ring, two blocks per CU (short-K linears) */
#define VMV_TILE_G128x160 8
#define VMV_TILE_P256x128 9   /* persistent 8-wave LDS-DMA kernel: ring kept full across tiles (gemm_pglds.hip) */
#define VMV_TILE_P256x160 10
#define VMV_TILE_PP256x128 11  /* 8-wave LDS-DMA kernel, ping-pong wave schedule */
#define VMV_TILE_PP256x160 12
#define VMV_TILE_Q128x128 13   /* persistent, 4 waves, 2-stage ring, TWO blocks per CU (gemm_pglds.hip) */
#define VMV_TILE_Q96x160  14
#define VMV_TILE_S256x128 15   /* persistent, wave-specialised: 8 MFMA waves + 4 LDS-DMA loader waves (gemm_sglds.hip) */
#define VMV_TILE_S192x160 16
#define VMV_TILE_S256x160 17
#define VMV_TILE_A128x160 18   /* A-stationary persistent kernel, deferred epilogue: K <= 320, wide N (gemm_astat.hip) */
#define VMV_TILE_A128x128 19
#define VMV_TILE_X256x320 20   /* 8 waves x 64 x {160,128,64} wave tiles, four-stage ring of 32-deep chunks (gemm_xglds.hip): the
                                  long-K convolutions / temporal convolutions of the large levels; the 256 x 256 form also carries the folded
                                  LayerNorm (rowstat) and GEGLU epilogues; all three accept ksplit > 1 (plain epilogue, even splits of >= 4
                                  32-deep chunks: VMV_EINVAL otherwise) */
#define VMV_TILE_X256x256 21
#define VMV_TILE_X256x128 22
#define VMV_TILE_RS       23   /* row-stationary kernel (gemm_rs.hip): the wave's rows of A live in registers for the whole K = 320 / 640
                                  range, W streams through an LDS ring, outputs leave per 32-column pair; rows per wave and the column
                                  split picked by the launcher — the short-K linears of the two large UNet levels */
#define VMV_TILE_RS512    24   /* the same, forced to 64 rows per wave (512-row blocks; K = 320 only) */
#define VMV_TILE_RS256    25   /* the same, forced to 32 rows per wave (256-row blocks) */
#define VMV_TILE_HALO     26   /* halo-resident 3 x 3 convolution for N <= 8 output channels (conv_halo.hip): 4 x 16 pixel tiles, the 6 x 18 halo and
                                  the weights in LDS — the VAE / UNet output heads */
#define VMV_TILE_TFR      27   /* frame-resident temporal convolution (gemm_tfr.hip): a block owns all F frames (12 <= F <= 24) of 192 / F pixels x 320
                                  output channels, A staged once through registers (optional GroupNorm + SiLU on the way, gn_table / gn_silu), the
                                  three taps as row-shifted views of one LDS tile; N % 320 == 0, C % 64 == 0 */

#define VMV_TILE_TQA      28   /* q | k | v projection + temporal attention (gemm_tqa.hip, VMV_EPI_TATTN only): a wave keeps all F frames of 48 / F
                                  pixels x K = 320 in registers, W streams head-major through the LDS ring, the head's 24 x 24 attention is finished
                                  in registers; 48 % F == 0, N = 192 * heads <= 3840, optional folded LayerNorm (colsum + ln_eps) */
#define VMV_TILE_W256x256 29   /* wide-wave register-staged kernel (gemm_wreg.hip): 4 waves x 128 x 128 of a 256 x 256 tile, accumulators in AGPRs,
                                  global -> registers -> LDS; one plain linear segment, K % 32 == 0.  EXPERIMENT (make EXPERIMENTS=1; forced tile only): correct,
                                  0.69 x of the 8-wave wide tile — DESIGN.md 10 */

#define VMV_TILE_X512x128 30   /* (ABI 11) gemm_xglds.hip with an 8 x 1 wave grid: 512 x 128 tile of 64 x 128 wave tiles — the N = 128 convolutions over
                                  millions of rows (the VAE's first level), where every 64 x 64-wave-tile kernel sits at 670 TFLOP/s; plain
                                  epilogue, no split-K */
#define VMV_TILE_Y256x128 31   /* (ABI 11) gemm_xglds.hip in 256-thread blocks: 4 x 1 waves of 64 x 128, 256 x 128 tile, three-stage ring, TWO blocks per CU
                                  (one's fill / epilogue under the other's main loop); plain, folded-LayerNorm and GEGLU epilogues; no split-K.  EXPERIMENT
                                  (make EXPERIMENTS=1; forced tile only): correct, 0.62-0.92 x of the one-block forms — DESIGN.md 10 */

int vmv_gemm(const VmvGemmParams* p, void* stream);
/* 1 if the host should record ONE VMV_EPI_TATTN launch for *p (a fused q | k | v + temporal-attention GEMM, epilogue already set)
 * instead of the q | k | v GEMM + vmv_attention pair: the fused kernel supports the shape and its grid fills the chip */
int vmv_gemm_tqa_ok(const VmvGemmParams* p);
/* 1 if vmv_gemm accepts *p (rowstat ignored) with in-loop LayerNorm statistics (VmvGemmParams.ln_eps), else 0 */
int vmv_gemm_ln_inline_ok(const VmvGemmParams* p);
/* 1 if vmv_gemm would run *p (tile = VMV_TILE_AUTO) on the row-stationary kernel, which takes the statistics of a folded
 * LayerNorm from its resident rows: the host then passes colsum + ln_eps and no rowstat (no statistics launch at all) */
int vmv_gemm_rs_ok(const VmvGemmParams* p);
/* 1 if vmv_gemm would run *p (tile = VMV_TILE_AUTO, a temporal convolution, with or without gn_table) on the frame-resident kernel
 * (VMV_TILE_TFR): the host then records statistics + vmv_groupnorm_table + this GEMM instead of statistics + apply + GEMM */
int vmv_gemm_tfr_ok(const VmvGemmParams* p);
/* the VMV_TILE_* configuration vmv_gemm's policy picks for *p when p->tile == VMV_TILE_AUTO (p->tile otherwise); host logic only:
 * no launch, no device access (a launcher may still fall back when it cannot address the operands) */
int vmv_gemm_pick_tile(const VmvGemmParams* p);
/* Everything vmv_gemm(p, stream) does on the host — argument validation, the tile policy (or the forced p->tile / p->ksplit), the chosen
 * launcher's own eligibility checks — without touching the device: VMV_OK iff the same call would launch a kernel, the VMV_E* code it
 * would return otherwise.  Needs no GPU.  The Python host calls it when a measured (tile, split-K) entry of tuned_gemm.json is about
 * to be forced on a recorded launch: a stale entry is dropped there, with a warning, instead of aborting the first replay. */
int vmv_gemm_validate(const VmvGemmParams* p);

/* ------------------------------------------------------------------------------------------------------
 * FeedForward of a BasicTransformerBlock in one launch (util.py:536-540 `x = ff(norm3(x)) + x`, FeedForward :553-578,
 * GEGLU :541-550), for C = 320 (the UNet's largest level):
 *     out = residual + W2 . ( (W1x LN(x) + b1x) * gelu(W1g LN(x) + b1g) ) + b2
 * The 4C-wide hidden activation stays in registers (csrc/gemm_ff.hip); rounding points are those of the two-GEMM form
 * (hidden rounded to elem before the down projection).  W1 / b1: the GEGLU up projection packed as for vmv_gemm with
 * VMV_EPI_GEGLU (x | gate rows interleaved in 16-row blocks; LayerNorm gamma / beta folded in when ln_eps > 0, as for
 * VmvGemmParams.colsum); W2: elem [C][4C] whose K axis is permuted inside every block of 32 channels to
 * [0-3, 16-19, 4-7, 20-23, 8-11, 24-27, 12-15, 28-31] (position -> channel), the order in which the kernel's MFMA outputs form
 * the next MFMA's operand.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t M, C;            /* rows; channels (320)                                               */
    const void* x; int32_t ldx;          /* elem rows [M][ldx]                                      */
    int32_t _pad0;
    const void* w1;          /* elem [8C][C]                                                       */
    const float* b1;         /* [8C] or NULL                                                       */
    const void* w2;          /* elem [C][4C], K-permuted (see above)                               */
    const float* b2;         /* [C] or NULL                                                        */
    const void* residual; int32_t ldr;   /* optional elem [M][ldr] (may alias x)                    */
    float ln_eps;            /* > 0: LayerNorm (no affine: folded) of each row of x first; 0: x as is */
    void* out; int32_t ldo;  /* elem [M][ldo]                                                      */
    int32_t _pad1;
} VmvFfParams;
int vmv_ff_fused(const VmvFfParams* p, void* stream);
/* 1 if vmv_ff_fused serves *p (host logic only) */
int vmv_ff_fused_ok(const VmvFfParams* p);

/* ------------------------------------------------------------------------------------------------------
 * GroupNorm(32 groups) over row blocks + optional SiLU (torch group_norm + silu: util.py:329,649,673,1014,
 * 1358-1373; unet_t2v.py:262; autoencoder.py Normalize).  A "stat group" is `rows_per_stat` consecutive rows:
 * H*W rows for the per-frame 4-D norms, F*H*W rows for the 5-D norms whose statistics span all frames
 * (SURVEY F9).  Two launches: partial sums (deterministic: fixed order or integer atomics), then normalise(+SiLU).
 * ---------------------------------------------------------------------------------------------------- */
typedef struct {
    const void* x;         /* elem [rows][ld]   (second source x1 optional: channels [C0, C0+C1) )  */
    const void* x1;
    int32_t ld, ld1;
    int32_t C0, C1;        /* C = C0 + C1, C % 32 == 0, (C/32) % 2 == 0, C0 % 8 == 0               */
    int32_t rows, rows_per_stat;
    int32_t chunk_rows;    /* rows per partial-sum block (rows_per_stat % chunk_rows may be != 0)   */
    float* partial;        /* workspace [nstat][nchunk][32][2] floats                              */
    const float* gamma;    /* [C] */
    const float* beta;     /* [C] */
    float eps;
    int32_t silu;          /* 1: y = silu(gn(x)) */
    void* y;               /* elem [rows][ldy] */
    int32_t ldy;
    int32_t fold_ranks;    /* 0/1: `partial` holds this launch's sums.  R > 1 (frame-sharded 5-D norms): apply
                              folds `partial` = [R][nstat][nchunk][32][2] — the all-gathered sums of R equally sized shards of
                              each stat group — and normalises by R * rows_per_stat rows.  Ignored by _stats.            */
    /* Statistics are taken of x - pilot, the pilot of a (stat group, channel group) being its first element (first row of the stat
     * group, first channel of the group): shift-invariant, no E[x^2] - mean^2 cancellation for |mean| >> sigma.
     * Optional stat-group totals (long stat groups: the all-frame norms have up to 256 chunks, which every apply block would
     * otherwise re-fold).  With `totals` set, every _stats block ADDS its 32 (sum, sumsq) pairs to the records
     * totals[stat][32][VMV_GN_NREP][VMV_GN_REC] = { sum_hi, sumsq_hi, sum_lo, sumsq_lo, pilot (fp32 bits), 3 x pad } with 64-bit
     * integer atomics (chunk c adds into replica c % VMV_GN_NREP — atomics on one record serialise — replica 0 holds the pilot):
     * two-limb fixed point (integer limb + 2^-40 fraction limb: a block's fp32 partial sum is represented exactly for 1e-3-sized
     * and for 3e3-sized activations alike, |sum| < 2^63).  Integer addition commutes, so the result is bitwise reproducible
     * whatever the arrival order, and no fold launch and no release fence is needed.  The accumulators must be ZERO when _stats
     * starts.  _apply reads totals instead of partial ([nstat][32][NREP][REC], or [R][nstat][32][NREP][REC] gathered when
     * fold_ranks = R > 1: each rank's sums are relative to ITS pilot and are moved to rank 0's in fp64; fold_ranks > 1 requires
     * totals) and, when `totals_clear` is set, zeroes `clear_count` int64 entries there — the accumulators of the NEXT norm (two
     * buffers used alternately: the buffer being cleared was last read one norm ago).                                          */
    int64_t* totals;
    int64_t* totals_clear;
    int32_t clear_count;
    int32_t _pad;
} VmvGroupNormParams;

#define VMV_GN_REC 8
#define VMV_GN_NREP 8
int vmv_groupnorm_stats(const VmvGroupNormParams* p, void* stream);
/* The per-channel scale / shift of the norm instead of its output: p->y is an fp32 table [nstat][2][C] (scale[c] = rstd * gamma[c],
 * shift[c] = beta[c] - mean * scale[c]; silu must be 0) for VmvGemmParams.gn_table.  Same statistics inputs as _apply. */
int vmv_groupnorm_table(const VmvGroupNormParams* p, void* stream);
int vmv_groupnorm_apply(const VmvGroupNormParams* p, void* stream);
/* One-launch GroupNorm for stat groups that fit on chip: a block stages all rows_per_stat rows of `cols` channels (a
 * whole number of groups, cols % 8 == 0, rows_per_stat * cols * 2 <= VMV_GN_FUSED_BYTES) in LDS, computes the statistics
 * TWO-PASS (mean, then sum of squared deviations: no E[x^2] - mean^2 cancellation), applies them and writes y — one
 * launch and one read of x instead of stats (+ fold) + apply.  partial / totals / chunk_rows / fold_ranks are unused.
 * Returns VMV_ERANGE when the group does not fit (use the two-kernel form).  The small levels of the UNet, where the
 * two- / three-launch form is pure launch latency. */
#define VMV_GN_FUSED_BYTES 131072
int vmv_groupnorm_fused(const VmvGroupNormParams* p, int32_t cols, void* stream);

/* LayerNorm over the channel axis of [rows][C] elem (nn.LayerNorm, eps 1e-5: util.py:528-530) */
typedef struct {
    const void* x; int32_t ldx;
    void* y; int32_t ldy;
    const float* gamma; const float* beta;
    int32_t rows, C;       /* C % 8 == 0, C <= 2048 */
    float eps;
    int32_t _pad;
    float* stats_out;      /* optional fp32 [rows][2]: write (mean, rstd) per row INSTEAD of y (y, gamma, beta may be NULL):
                              the statistics pass of a LayerNorm folded into its consumer GEMM (VmvGemmParams.rowstat) */
} VmvLayerNormParams;
int vmv_layernorm(const VmvLayerNormParams* p, void* stream);

/* Row softmax: p[r][c] = softmax_c(scale * s[r][c]), fp32 scores -> elem probabilities.  Used by the VAE decoder's
 * single-head, 512-wide attention (AttnBlock, autoencoder.py:366-390), which runs as GEMM -> softmax -> GEMM. */
typedef struct {
    const float* s; int32_t lds;
    void* p; int32_t ldp;          /* elem */
    int32_t rows, n;               /* n % 4 == 0 */
    float scale;
    int32_t _pad;
} VmvSoftmaxParams;
int vmv_softmax_rows(const VmvSoftmaxParams* p, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Softmax attention, head_dim 64, no bias, scale given (xformers memory_efficient_attention at
 * util.py:253,258); optional causal mask (key index <= query index: the text tower of clip_embedder.py:192-201).  One kernel for the three uses, selected by the index maps:
 *   spatial self  : problems (b f, head), Nq = Nk = H*W
 *   spatial cross : same queries, Nk = 77 text tokens shared by all frames of a batch item (kv_div = F)
 *   temporal      : problems (b, pixel, head), Nq = Nk = F, rows strided by H*W   
 * Row address of sequence position i of problem `o`, head h:
 *     base + (o / inner) * s_outer + (o % inner) * s_inner + i * s_row + h * head_dim      (elements)
 * ---------------------------------------------------------------------------------------------------- */
typedef struct {
    int64_t s_outer, s_inner, s_row;
    int32_t inner; int32_t _pad;
} VmvSeqMap;

typedef struct {
    const void* q; const void* k; const void* v; void* o;   /* elem */
    VmvSeqMap qm, km, vm, om;
    int32_t n_outer;       /* number of q problems per head */
    int32_t kv_div;        /* kv problem index = o / kv_div */
    int32_t heads;
    int32_t Nq, Nk;
    float scale;
    int32_t head_dim;      /* 64 (0 = 64) or 32; 32 only for long sequences (LGM MVAttention, core/attention.py:67-84) */
    int32_t causal;        /* != 0: scores of keys j > query i are masked out (requires Nq == Nk, head_dim 64)         */
} VmvAttnParams;
int vmv_attention(const VmvAttnParams* p, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Sampler glue kernels (DiffusionDDIM.p_mean_variance / ddim_sample: diffusion_ddim.py:157-160,192-195,233-243;
 * layout moves of unet_t2v.py:348,368; embeddings unet_t2v.py:326-335).
 * ---------------------------------------------------------------------------------------------------- */
/* latent [nb_src][C][F][H][W] fp32 -> rows [nrep*nb_src*F*H*W][Cpad] elem (channels zero-padded to Cpad, the
 * whole volume written nrep times: the cond / uncond CFG branches see the same x_t) */
int vmv_latent_to_rows(const float* x, void* rows, int nb_src, int C, int F, int H, int W, int Cpad, int nrep,
                       void* stream);
/* same gather, but ONLY channels [0, C) of each ld-wide row are written (the rest of the row is left untouched): the
 * I2VGen input rows carry x_t in channels 0..3 and the step-invariant image `concat` in 4..7 (unet_i2vgen.py:383) */
int vmv_latent_to_rows_keep(const float* x, void* rows, int nb, int C, int F, int H, int W, int ld, int nrep, void* stream);
/* rows [n*HW][ld] (elem or fp32) -> image/latent [n][C][H][W] fp32 (C <= ld) */
int vmv_rows_to_nchw(const void* rows, int rows_fp32, int ld, float* out, int n, int C, int HW, void* stream);

typedef struct {
    const float* eps_rows;   /* fp32 [2][F*H*W][ld]: branch 0 = conditional, 1 = unconditional        */
    int32_t ld;
    int32_t C, F, HW;
    float guide_scale;
    float c_recip, c_recipm1;      /* sqrt(1/abar_t), sqrt(1/abar_t - 1)   (eps-prediction)            */
    float c_sqrt_ac, c_sqrt_1mac;  /* sqrt(abar_t), sqrt(1-abar_t)         (v-prediction)              */
    float a_prev;                  /* abar_{t-stride}                                                  */
    int32_t v_pred;                /* 0: eps-prediction, 1: v-prediction                               */
    float* xt;                     /* fp32 [C][F][HW] updated in place to x_{t-1}                      */
    float* x0_out;                 /* optional fp32 [C][F][HW] predicted x0                            */
    /* Sampler options of the reference signature (diffusion_ddim.py:201-205, 233-243).  clamp > 0: x0 is restricted to
     * [-clamp, clamp] before eps is re-derived from it; sigma > 0 (stochastic DDIM, eta > 0; the host computes
     * sigma = eta * sqrt((1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)) in fp32 as the reference does): the update becomes
     * sqrt(a_prev) x0 + sqrt(1 - a_prev - sigma^2) eps + sigma * noise with noise = fp32 [C][F][HW] standard normal (required then). */
    float clamp;
    float sigma;
    const float* noise;
} VmvDdimParams;
int vmv_cfg_ddim_step(const VmvDdimParams* p, void* stream);

/* DiagonalGaussianDistribution.sample() * scale (autoencoder.py:213-226, get_first_stage_encoding :19-28):
 * moments rows fp32 [n*HW][ld] = (mean[zc] | logvar[zc]); z[n][zc][HW] = scale * (mean + exp(0.5*clamp(logvar,-30,20)) * noise) */
int vmv_posterior_sample(const float* moments_rows, int ld, const float* noise, float* z, int n, int zc, int HW, float scale,
                         void* stream);
/* e[r][c] = silu(temb[(r / rows_per_t)][c] + (cam ? cam[r % cam_rows][c] : 0)) -> elem [rows][C] */
int vmv_emb_combine_silu(const float* temb, const float* cam, void* out, int rows, int C, int rows_per_t,
                         int cam_rows, void* stream);
/* sinusoidal timestep embedding (cos || sin), out fp32->elem [n][dim]  (util.py:177-189) */
int vmv_sinusoidal(const float* t, void* out_elem, int n, int dim, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * I2VGen-XL front-end helpers (once per sample; unet_i2vgen.py:331-346, 156-162).
 * ---------------------------------------------------------------------------------------------------- */
/* TransformerV2(depth 1, dim 4, heads 2, dim_head 4, mlp 16) over the F frames of every pixel (util.py:1091-1148):
 *   x = to_out(softmax(q k^T / 2) v) + x  with q,k,v = to_qkv(LayerNorm(x));   x = W2 gelu(W1 x + b1) + b2 + x
 * in: elem rows [F*HW][ld_in] (4 channels used); out: `scale` * result written to elem rows [nrep][F*HW][ld_out] at
 * channel offset 0 of `out` (pass out = base + 4 to fill channels 4..7).  w = packed fp32 parameter block:
 *   ln_g[4] ln_b[4] Wqkv[24][4] Wo[4][8] bo[4] W1[16][4] b1[16] W2[4][16] b2[4]   (288 floats). */
int vmv_i2v_temporal_adapter(const void* in, int ld_in, void* out, int ld_out, const float* w, int F, int HW, int nrep,
                             float scale, void* stream);
/* nn.AdaptiveAvgPool2d((OH,OW)) on channels-last rows: in [n*IH*IW][ld] -> out [n*OH*OW][ldo], C % 8 == 0 */
int vmv_adaptive_avgpool_rows(const void* in, int ld, void* out, int ldo, int n, int C, int IH, int IW, int OH, int OW,
                              void* stream);

/* Glue of the LGM-refined sampling step (unet_t2v.py:404-433, diffusion_ddim.py:179-182,224-243).
 * vmv_lgm_x0_views: z[v] = inv_scale * (c_recip * xt[:, :, idx[v]] - c_recipm1 * eps_branch[:, :, idx[v]]) for 4 views;
 *   eps_rows fp32 [2*F*HW][ld] (branch-major), xt [1][C][F][HW], out [4][C][HW].
 * vmv_lgm_pack_input: decoded VAE images [4][3][HW] in [-1,1] -> clamp(0.5*d+0.5, 0, 1) -> (x - mean)/std (ImageNet) into
 *   out[:, 0:3], rays [4][6][HW] copied into out[:, 3:9]; out [4][9][HW].
 * vmv_lgm_render_to_vae: rendered images [V][3][S_in][S_in] in [0,1] -> nearest resampling to [S][S] (F.interpolate(...,
 *   (S, S), mode='nearest'): source index floor(dst * S_in / S), unet_t2v.py:425-427) and
 *   (x - 0.5) / 0.5; out [V][3][S][S].
 * vmv_ddim_x0_step: x0 = u + guide * (c - u) (CFG on the two branches' latent_z), eps = (c_recip*xt - x0)/c_recipm1,
 *   x0 clamped to +-clamp when clamp > 0 (diffusion_ddim.py:204-205), xt <- sqrt(a_prev) * x0 + sqrt(1 - a_prev - sigma^2) * eps
 *   + sigma * noise (:233-243; noise may be NULL when sigma == 0), in place; all [n] floats.  (ABI 10: clamp / sigma / noise) */
int vmv_lgm_x0_views(const float* eps_rows, int ld, int branch, const float* xt, int C, int F, int HW, const int32_t* idx4,
                     float c_recip, float c_recipm1, float inv_scale, float* out, void* stream);
int vmv_lgm_pack_input(const float* decoded, const float* rays, float* out, int nviews, int HW, void* stream);
int vmv_lgm_render_to_vae(const float* images, float* out, int nviews, int S_in, int S, void* stream);
int vmv_ddim_x0_step(const float* x0_cond, const float* x0_uncond, float* xt, long n, float guide, float c_recip,
                     float c_recipm1, float a_prev, float clamp, float sigma, const float* noise, void* stream);

/* LGM Gaussian activations (core/models.py:37-43,102-112): raw fp32 rows [n][ld >= 14] -> out [n][14] =
 * (pos.clamp(-1,1) x3, sigmoid(opacity), 0.1*softplus(scale) x3, rotation x4, 0.5*tanh(rgb)+0.5 x3).  The reference applies
 * F.normalize with its default dim=1 to the [B, N, 4] rotation block, i.e. each quaternion COMPONENT is divided by
 * max(L2 norm over the n Gaussians, 1e-12) — reproduced as is.  workspace: >= 1024 floats (deterministic two-pass sum). */
int vmv_gaussian_activation(const float* raw, int ld, float* out, int n, float* workspace, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Forward Gaussian-splatting rasteriser of the LGM refinement branch.  Replaces the reference's use of the third-party
 * extension diff_gaussian_rasterization (core/gs.py:7-10,57-83: GaussianRasterizationSettings(sh_degree=0, bg, tanfov,
 * viewmatrix, projmatrix) + rasterizer(means3D, colors_precomp, opacities, scales, rotations)); one view per call pair.
 * All buffers are caller-owned device memory.  Two calls because the number of (tile, Gaussian) instances is data
 * dependent: vmv_gs_preprocess fills the per-Gaussian arrays and `offsets` (inclusive scan of tiles_touched; the host
 * reads offsets[N-1] = num_rendered and sizes keys/vals), vmv_gs_render bins, sorts (rocprim radix sort), and blends.
 * out_color [3][size][size] is clamped to [0,1] (core/gs.py:84), out_alpha [size][size] (optional) = sum alpha*T.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct {
    const float* gaussians;      /* [N][14]: pos 3, opacity 1, scale 3, rotation (r,x,y,z) 4 — used as given —, rgb 3 */
    int32_t N;
    int32_t size;                /* square image, pixels */
    const float* view;           /* [16] row-major 4x4, row-vector convention (p @ M): cam_view        */
    const float* view_proj;      /* [16]                                            : cam_view_proj   */
    float tan_half_fov;
    float bg[3];
    float* depth;                /* [N]   preprocess outputs */
    float* xy;                   /* [N][2] */
    float* conic_opacity;        /* [N][4] */
    int32_t* rect;               /* [N][4] tile rectangle x0,y0,x1,y1 */
    uint32_t* tiles_touched;     /* [N] */
    uint32_t* offsets;           /* [N] inclusive scan */
    void* scan_temp; size_t scan_temp_bytes;
    uint64_t* keys; uint64_t* keys_sorted;      /* [num_rendered] */
    uint32_t* vals; uint32_t* vals_sorted;      /* [num_rendered] */
    int32_t num_rendered; int32_t _pad;
    void* sort_temp; size_t sort_temp_bytes;
    uint32_t* ranges;            /* [tiles][2] */
    float* out_color; float* out_alpha;
} VmvGsParams;
int vmv_gs_workspace_bytes(int n_gaussians, int n_instances, size_t* scan_bytes, size_t* sort_bytes);
int vmv_gs_preprocess(const VmvGsParams* p, void* stream);
int vmv_gs_render(const VmvGsParams* p, void* stream);

/* The same rasteriser for ALL views of a call in one pass (the LGM branch renders 24 views of each CFG branch's Gaussians per refined
 * step, core/gs.py:41-83 loops `for b in range(B): for v in range(V):` around the extension): "view" vv = b * V + v uses sample b's
 * Gaussians and the vv-th camera.  ONE preprocess launch over (view, Gaussian); the B * V * N records are ranked by (view, depth) with a
 * stable 64-bit radix sort (1.6 M pairs at the VideoMV size) and their tile counts scanned in rank order, so the instances are emitted
 * depth-ordered per view; ONE radix sort of every instance on its 32-bit (view, tile) id vv * tiles + tile ALONE (round 6: two 8-bit
 * passes over 8-byte pairs at 24 views instead of six over 12-byte pairs with the depth in the key) — stable, so a tile's instances
 * stay in (depth, Gaussian index) order, the order the per-view entry points' 64-bit (tile, depth) sort produces — one ranges and one
 * blend launch (grid = tiles x views; a tile without instances writes the background): ONE host read of the instance total per call
 * instead of one host round trip per view.  Per-(view, Gaussian) arrays hold B * V * N entries; keys / vals num_rendered (keys: the
 * 32-bit ids use the first half of the 8 bytes per instance).  scan_temp is the preprocess workspace (ranking buffers + primitive
 * temporaries, sized by vmv_gs_batch_workspace_bytes) and must stay untouched between _preprocess and _render; `offsets` is the scan
 * in RANK order (offsets[B*V*N - 1] = num_rendered).  Shared blend code, same order: the images are bit-identical to the per-view path.
 *   vmv_gs_batch_key_bits(n_views, size)       -> 32 + ceil(log2(n_views * tiles)) (ABI-stable; the sort uses the low part)
 *   vmv_gs_batch_workspace_bytes(n_view_gaussians, n_instances, key_bits, &scan, &sort)
 *   vmv_gs_batch_preprocess -> offsets[B*V*N - 1] = num_rendered (host reads it, sizes keys / vals / sort_temp), vmv_gs_batch_render. */
typedef struct {
    const float* gaussians;      /* [B][N][14] (layout of VmvGsParams.gaussians per sample)                     */
    int32_t B, N, V, size;
    const float* views;          /* [B * V][16] cam_view, row-major, row-vector convention                     */
    const float* view_projs;     /* [B * V][16] cam_view_proj                                                  */
    float tan_half_fov;
    float bg[3];
    float* depth;                /* [B * V * N]   preprocess outputs, view-major */
    float* xy;                   /* [B * V * N][2] */
    float* conic_opacity;        /* [B * V * N][4] */
    int32_t* rect;               /* [B * V * N][4] */
    uint32_t* tiles_touched;     /* [B * V * N] */
    uint32_t* offsets;           /* [B * V * N] inclusive scan over all views */
    void* scan_temp; size_t scan_temp_bytes;
    uint64_t* keys; uint64_t* keys_sorted;      /* [num_rendered] */
    uint32_t* vals; uint32_t* vals_sorted;      /* [num_rendered] Gaussian index inside its sample */
    int32_t num_rendered; int32_t _pad;
    void* sort_temp; size_t sort_temp_bytes;
    uint32_t* ranges;            /* [B * V * tiles][2] */
    float* out_color;            /* [B * V][3][size][size], clamped to [0, 1] */
    float* out_alpha;            /* [B * V][size][size] or NULL */
} VmvGsBatchParams;
int vmv_gs_batch_key_bits(int n_views, int size);
int vmv_gs_batch_workspace_bytes(int n_view_gaussians, int n_instances, int key_bits, size_t* scan_bytes, size_t* sort_bytes);
int vmv_gs_batch_preprocess(const VmvGsBatchParams* p, void* stream);
int vmv_gs_batch_render(const VmvGsBatchParams* p, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Block permute-copy (frame-sharded sampling, DESIGN.md §8: packs / unpacks the all-to-all buffers that switch an
 * activation between the frame-major shard [B][F/R][HW][C] and the pixel-major shard [B][F][HW/R][C]; the reference has
 * no counterpart — its multi-GPU mode is replicas only, inference_text2video_entrance.py:152-156).
 * dst[i0][i1][i2][0..inner) = src[i0*ss0 + i1*ss1 + i2*ss2 + (0..inner)], dst contiguous; all counts in 16-byte units.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct {
    const void* src;
    void* dst;
    int32_t n0, n1, n2;
    int32_t inner16;              /* contiguous 16-byte vectors per (i0,i1,i2) block */
    int64_t ss0, ss1, ss2;        /* source strides, 16-byte units */
} VmvCopyParams;
int vmv_permute_copy(const VmvCopyParams* p, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Collectives of the frame-sharded sampler (SURVEY §8b "vmv_comm_* wrappers, RCCL communicator handle injected
 * from Python", §8e).  The reference has no counterpart: its multi-GPU mode is replicas (inference_text2video_entrance.py:79,
 * 152-156).  One process per GPU; the host creates the communicator ONCE — rank 0 draws an id with vmv_comm_unique_id(), ships its
 * 128 bytes to the other ranks by any means (the Python host broadcasts it over torch.distributed), every rank calls
 * vmv_comm_create() — and records VMV_OP_COMM ops into its plans, so the 183 collectives of a branch forward are issued from the
 * C replay loop on the replay's stream (and captured into its hipGraph) instead of from Python between plan segments.
 * RCCL is resolved at run time: vmv_comm_load(path) dlopen()s the librccl.so the process already uses (path NULL / "" =
 * "librccl.so" through the loader's search path); libvmv itself does not link against it.
 *   VMV_COMM_ALL_TO_ALL : send = [world][bytes] (chunk j goes to rank j), recv = [world][bytes] (chunk i came from rank i) — the
 *                         frame-major <-> pixel-major layout switch;
 *   VMV_COMM_ALL_GATHER : send = [bytes], recv = [world][bytes] — the GroupNorm totals of the all-frame norms, K | V of the
 *                         north-star temporal-attention form.
 * vmv_comm_create_sim(world, rank): a communicator whose W - 1 peers are absent — every collective becomes a device-local copy of
 * the same byte count (all-to-all: recv = send; all-gather: recv[j] = send for all j), so the rank-local plan of a W-GPU run can be
 * replayed and timed on ONE GPU (bench.py --simulate-rank).  Its output is not a sample.
 * ---------------------------------------------------------------------------------------------------- */
#define VMV_COMM_ALL_TO_ALL 0
#define VMV_COMM_ALL_GATHER 1
#define VMV_COMM_ID_BYTES   128
typedef struct VmvComm VmvComm;
typedef struct {
    const VmvComm* comm;
    int32_t kind;            /* VMV_COMM_*                                                          */
    int32_t _pad;
    const void* send;
    void* recv;
    int64_t bytes;           /* per-rank chunk, > 0                                                 */
} VmvCommParams;
int      vmv_comm_load(const char* rccl_path);
int      vmv_comm_loaded(void);
int      vmv_comm_unique_id(void* id128);                               /* rank 0: VMV_COMM_ID_BYTES bytes out */
VmvComm* vmv_comm_create(const void* id128, int world, int rank);       /* collective over the `world` ranks; NULL on failure */
VmvComm* vmv_comm_create_sim(int world, int rank);
void     vmv_comm_destroy(VmvComm* comm);
int      vmv_comm_world(const VmvComm* comm);
int      vmv_comm_rank(const VmvComm* comm);
int      vmv_comm_is_sim(const VmvComm* comm);
int      vmv_comm_run(const VmvCommParams* p, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Plan: a recorded sequence of the launches above, replayed with one call (host-side launch overhead of
 * >1000 kernels per forward would otherwise dominate).
 * ---------------------------------------------------------------------------------------------------- */
typedef struct VmvPlan VmvPlan;
#define VMV_OP_GEMM        1
#define VMV_OP_GN_STATS    2
#define VMV_OP_GN_APPLY    3
#define VMV_OP_LAYERNORM   4
#define VMV_OP_ATTENTION   5
#define VMV_OP_SOFTMAX     6
#define VMV_OP_COPY        7
#define VMV_OP_GN_FUSED    8   /* args: VmvGroupNormParams with chunk_rows = cols of vmv_groupnorm_fused */
#define VMV_OP_FF          9   /* args: VmvFfParams */
#define VMV_OP_GN_TABLE    10  /* args: VmvGroupNormParams, y = the fp32 table (vmv_groupnorm_table) */
#define VMV_OP_COMM        11  /* args: VmvCommParams — a collective of the frame-sharded sampler, issued on the replay's stream */
VmvPlan* vmv_plan_create(void);
void     vmv_plan_destroy(VmvPlan* plan);
int      vmv_plan_add(VmvPlan* plan, int op, const void* params, size_t nbytes);
int      vmv_plan_size(const VmvPlan* plan);
int      vmv_plan_run(const VmvPlan* plan, void* stream);
/* run ops [first, last) only — used by tests and profiling */
int      vmv_plan_run_range(const VmvPlan* plan, int first, int last, void* stream);
/* The plan as a hipGraph: vmv_plan_capture() records ONE replay of the whole plan on `stream` in stream-capture mode and
 * instantiates it (nothing executes; every launcher takes the stream explicitly and allocates nothing, so all of them — the RCCL
 * collectives of VMV_OP_COMM included — are capturable); vmv_graph_launch() enqueues the instantiated graph: one host call and one
 * submission per forward instead of ~800 launches.  Run the plan once eagerly first (kernels set their LDS-size attribute on first
 * use).  The argument blocks are baked in: re-capture after editing the plan.  Returns NULL / an error code on failure. */
typedef struct VmvGraph VmvGraph;
VmvGraph* vmv_plan_capture(const VmvPlan* plan, void* stream);
int       vmv_graph_launch(const VmvGraph* graph, void* stream);
int       vmv_graph_nodes(const VmvGraph* graph);
void      vmv_graph_destroy(VmvGraph* graph);

#ifdef __cplusplus
}
#endif
#endif /* VMV_H */

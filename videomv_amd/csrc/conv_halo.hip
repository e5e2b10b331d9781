// conv_halo.hip — HALO-RESIDENT 3 x 3 convolution for FEW output channels (N <= 8): the VAE's conv_out (128 -> 3, autoencoder.py
// Decoder.conv_out), the encoder's conv_out (512 -> 8) and the UNet's eps head (320 -> 4, unet_t2v.py:265 `self.out`).
//
// Why these and only these.  The implicit-GEMM kernels fetch the A tile of every tap again (9 x the activation bytes through the
// CU), which is what bounds them (FLOPs per LDS-DMA byte x ~8 TB/s).  With N = 128 .. 1280 output channels keeping the
// activations on chip does not pay: the halo tile limits a block to ~128 - 256 pixels and W then has to be re-streamed per tile.  With
// N <= 8 the weights are 9 - 18 KB and simply live in LDS next to the tile, and the tile kernels were worst here: the narrowest
// GEMM tile is 64 columns wide, so 16 x the MFMA work and the full 9 x activation traffic for 3 useful columns — conv_out of the
// VAE ran at 35 TFLOP/s, 1.0 ms for a 1-GB read; here 369 us (same box), the encoder's head 74 -> 30 us.
//
// Structure.  A block owns a 4 x 16 pixel tile of one image.  Per chunk of <= 128 input channels its 6 x 18 halo (108 pixels x
// 256 B = 27 KB) arrives by LDS-DMA — every input pixel is fetched once per tile (1.7 x the tensor instead of 9 x); pixels outside
// the image are out-of-range lanes of the buffer descriptor and arrive as zeros (= the convolution's padding) — the 16-byte slot of
// pixel P is XOR-swizzled by P so that the 16 consecutive pixels of a fragment read hit 16 different bank groups.  The chunk's
// weights ([tap][k-step][row < 4 or 8][k-quarter], 16-byte pieces) sit behind it.  Wave w computes image row w of the
// tile: per (tap, 32-channel k-step) one v_mfma_f32_16x16x32 with A = the weight fragment (rows >= N are zero lanes, never
// read from LDS) and B = 16 pixels x 32 channels read from the halo tile at the tap's offset.  The accumulator layout gives lane
// (pixel = lane & 15, lane >> 4 == 0) output channels 0 .. 3 of its pixel: one 16-byte store per pixel for the fp32 image.
#include "gemm_glds_common.h"

using namespace vmvg;

namespace {

#ifndef VMV_CH_TH
#define VMV_CH_TH 4            // tile height (image rows): 4 (one row per wave; 36-KB blocks, four per CU) or 8 (two rows; 55 KB, two per CU).
                               // Measured at the VAE's head (24 x 320 x 512 x 128 -> 3): 369 us against 484 — the tile is bound by its DMA phase
                               // (issue + latency), which more resident blocks hide, not by the 1.7 x instead of 1.4 x halo bytes
#endif
constexpr int CH_TH = VMV_CH_TH, CH_TW = 16, CH_HH = CH_TH + 2, CH_HW = CH_TW + 2, CH_HPX = CH_HH * CH_HW;      // 180 (108) halo pixels
constexpr int CH_RPW = CH_TH / 4;                              // image rows per wave
constexpr int CH_CMAX = 128;                                   // input channels per staged chunk
constexpr int CH_HALO_BYTES = ((CH_HPX + 3) / 4) * 4 * CH_CMAX * 2;      // rounded up to whole 1-KB DMA instructions (4 pixels each)
constexpr int ch_w_bytes(int Nr) { return 9 * (CH_CMAX / 32) * Nr * 4 * 16; }     // [tap][k-step][Nr rows][4 k-quarters] x 16 B: 9 / 18 KB
static_assert(CH_TH == 8 || CH_TH == 4, "tile height");

__global__ __launch_bounds__(256) void conv_halo_kernel(const VmvGemmParams p, const int C, const int tiles_x, const int tiles_y) {
    VMV_KERNEL_ENTER();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fgrp = lane >> 4;
    const int H = p.OH, W = p.OW;
    const int tpi = tiles_x * tiles_y;
    const int img = blockIdx.x / tpi, t = blockIdx.x - img * tpi;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int y0 = ty * CH_TH, x0 = tx * CH_TW;
    const int Nr = p.N <= 4 ? 4 : 8;                           // weight rows kept in LDS
    const VmvGemmSeg& sg = p.seg[0];
    const int ld = sg.ld;
    const long img_row0 = (long)img * H * W;
    // one descriptor per image: 32-bit byte offsets inside it (launcher: H * W * ld * 2 < 2^31)
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(sg.src) + img_row0 * ld * 2), 0, (uint32_t)H * (uint32_t)W * (uint32_t)ld * 2u, SRD_FLAGS);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, (uint32_t)Nr * (uint32_t)p.ktot * 2u, SRD_FLAGS);
    unsigned char* halo = smem;
    unsigned char* wlds = smem + CH_HALO_BYTES;

    f32x4_t acc[CH_RPW];
#pragma unroll
    for (int r = 0; r < CH_RPW; ++r) acc[r] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < C; c0 += CH_CMAX) {
        const int cc = (C - c0) < CH_CMAX ? (C - c0) : CH_CMAX;            // 128, 64 or 32 (launcher: C % 32 == 0, tail a power of two)
        const int SL = cc >> 3;                                            // 16-byte slots per pixel: 16 / 8 / 4
        const int slog = SL == 16 ? 4 : SL == 8 ? 3 : 2;
        const int pshift = 4 - slog;                                       // swizzle key = (P >> pshift) & (SL - 1)
        const int KS = cc >> 5;                                            // 32-channel k-steps
        if (c0 > 0) __syncthreads();                                       // everyone is done reading the previous chunk
        // ---- halo tile: piece q = (pixel P, slot s'), lane-linear; the lane fetches slot s' ^ key(P) of source pixel P
        const int npiece = CH_HPX * SL;
        for (int q0 = wave * 64; q0 < npiece; q0 += 256) {
            const int q = q0 + lane;
            const int P = q >> slog, sp = q & (SL - 1);
            const int hy = P / CH_HW, hx = P - hy * CH_HW;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const int s = sp ^ ((P >> pshift) & (SL - 1));
            const bool ok = P < CH_HPX && y >= 0 && y < H && x >= 0 && x < W;
            const uint32_t vo = ok ? (uint32_t)((y * W + x) * ld + c0 + s * 8) * 2u : OOB;
            VMV_BLDS16(a_rsrc, halo + q0 * 16, vo, 0);
        }
        // ---- weights of the chunk: piece ((tap * KS + ks) * Nr + m) * 4 + kg  <-  W[m][tap * C + c0 + ks * 32 + kg * 8 .. + 8]
        const int nwp = 9 * KS * Nr * 4;
        for (int q0 = wave * 64; q0 < nwp; q0 += 256) {
            const int q = q0 + lane;
            const int kg = q & 3, m = (q >> 2) % Nr, rest = (q >> 2) / Nr;
            const int ks = rest % KS, tap = rest / KS;
            const uint32_t vo = q < nwp ? (uint32_t)(m * p.ktot + tap * C + c0 + ks * 32 + kg * 8) * 2u : OOB;
            VMV_BLDS16(w_rsrc, wlds + q0 * 16, vo, 0);
        }
        wait_vmcnt<0>();
        __syncthreads();
        // ---- 2 image rows of 16 pixels per wave
        const bool wl = frow < Nr;                                          // lanes that hold a real weight row
        // (fully unrolled, k-steps guarded by a uniform test: the scheduler can then run the fragment reads of the next steps under
        //  the current MFMAs; measured neutral at the VAE's head, 452 against 436 us: the tile is bound by its DMA phase, not by this loop)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;                      // halo offsets 0 .. 2 (= image offsets -1 .. 1)
#pragma unroll
            for (int ks = 0; ks < CH_CMAX / 32; ++ks) {
                if (ks < KS) {
                    u32x4_t wf = {0u, 0u, 0u, 0u};
                    if (wl) wf = *reinterpret_cast<const u32x4_t*>(wlds + ((((tap * KS + ks) * Nr + frow) << 2) + fgrp) * 16);
#pragma unroll
                    for (int r = 0; r < CH_RPW; ++r) {
                        const int P = (CH_RPW * wave + r + dy) * CH_HW + frow + dx;
                        const int slot = (ks * 4 + fgrp) ^ ((P >> pshift) & (SL - 1));
                        const u32x4_t af = *reinterpret_cast<const u32x4_t*>(halo + (P * SL + slot) * 16);
                        acc[r] = VMV_MFMA16(__builtin_bit_cast(elem8_t, wf), __builtin_bit_cast(elem8_t, af), acc[r], 0, 0, 0);
                    }
                }
            }
        }
    }
    // ---- epilogue: lane (pixel frow of the row, channels 4 fgrp .. + 3); bias, fp32 or 16-bit store
    const int nch = p.N - 4 * fgrp;                                         // channels this lane owns (<= 0: none)
    if (nch <= 0) return;
#pragma unroll
    for (int r = 0; r < CH_RPW; ++r) {
        const int y = y0 + CH_RPW * wave + r, x = x0 + frow;
        if (y >= H || x >= W) continue;
        f32x4_t v = acc[r];
        if (p.bias) {
            const float* b = p.bias + 4 * fgrp;
            v.x += b[0]; if (nch > 1) v.y += b[1]; if (nch > 2) v.z += b[2]; if (nch > 3) v.w += b[3];
        }
        act_apply(v, p.act);
        const long row = img_row0 + (long)y * W + x;
        if (p.out_fp32) {
            float* o = reinterpret_cast<float*>(p.out) + row * p.ldo + 4 * fgrp;
            if (nch >= 4 && (p.ldo & 3) == 0) *reinterpret_cast<f32x4_t*>(o) = v;
            else { o[0] = v.x; if (nch > 1) o[1] = v.y; if (nch > 2) o[2] = v.z; if (nch > 3) o[3] = v.w; }
        } else {
            uint16_t* o = reinterpret_cast<uint16_t*>(p.out) + row * p.ldo + 4 * fgrp;
            const uint32_t lo = pack_elem2(v.x, v.y), hi = pack_elem2(v.z, v.w);
            if (nch >= 4 && (p.ldo & 3) == 0) *reinterpret_cast<u32x2_t*>(o) = u32x2_t{lo, hi};
            else { o[0] = (uint16_t)lo; if (nch > 1) o[1] = (uint16_t)(lo >> 16); if (nch > 2) o[2] = (uint16_t)hi; if (nch > 3) o[3] = (uint16_t)(hi >> 16); }
        }
    }
}

}  // namespace

// 9 SPATIAL segments of ONE source in the tap order of ops.conv3x3_segs (dy-major, offsets -1 .. 1), stride 1, no up-sampling,
// N <= 8 output channels, no residual / row vector / GEGLU / split-K / folded norms.
bool vmv_conv_halo_supported(const VmvGemmParams& p) {
    if (p.nseg != 9 || p.N > 8 || p.N <= 0 || (p.N & 3) || p.stride != 1 || p.ups != 0) return false;
    if (p.OH <= 0 || p.OW <= 0 || p.IH != p.OH || p.IW != p.OW || p.M % (p.OH * p.OW)) return false;
    if (p.residual || p.rowvec || p.epilogue != VMV_EPI_NONE || p.ksplit > 1 || p.rowstat || p.colsum || p.wgroup_rows || p.gn_table) return false;
    const int C = p.seg[0].k;
    if (C <= 0 || (C & 31) || p.ktot != 9 * C) return false;
    const int tail = C % CH_CMAX;
    if (tail != 0 && tail != 64 && tail != 32) return false;              // the staged pixel is 16 / 8 / 4 slots wide
    for (int t = 0; t < 9; ++t) {
        const VmvGemmSeg& s = p.seg[t];
        if (s.mode != VMV_SEG_SPATIAL || s.src != p.seg[0].src || s.ld != p.seg[0].ld || s.k != C) return false;
        if (s.d0 != t / 3 - 1 || s.d1 != t % 3 - 1) return false;
    }
    if ((p.seg[0].ld & 7) || !vmv_aligned16(p.seg[0].src) || !vmv_aligned16(p.W)) return false;
    if ((long)p.OH * p.OW * p.seg[0].ld * 2 >= (1L << 31) - 65536) return false;
    if (p.out_fp32 ? (((uintptr_t)p.out) & 3) != 0 : (((uintptr_t)p.out) & 1) != 0) return false;
    if (p.ldo < p.N) return false;
    return true;
}

int vmv_conv_halo_launch(const VmvGemmParams& p, hipStream_t st) {
    if (!vmv_conv_halo_supported(p)) return VMV_GLDS_UNSUPPORTED;
    const int tiles_x = (p.OW + CH_TW - 1) / CH_TW, tiles_y = (p.OH + CH_TH - 1) / CH_TH;
    const long nimg = p.M / (p.OH * p.OW);
    const long blocks = nimg * tiles_x * tiles_y;
    if (blocks > 0x7fffffffL) return VMV_ERANGE;
    static std::atomic<unsigned long long> attr{0};
    if (const int rc_attr = vmv_lds_attr_once(attr, reinterpret_cast<const void*>(&conv_halo_kernel), CH_HALO_BYTES + ch_w_bytes(8))) return rc_attr;
    const int lds = CH_HALO_BYTES + ch_w_bytes(p.N <= 4 ? 4 : 8);          // 36 / 45 KB (TH = 4), 55 / 64 KB (TH = 8)
    VMV_LAUNCH(conv_halo_kernel, dim3((unsigned)blocks), dim3(256), lds, st, p, p.seg[0].k, tiles_x, tiles_y);
    return vmv_launch_status();
}

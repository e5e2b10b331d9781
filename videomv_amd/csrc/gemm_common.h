// gemm_common.h — pieces shared by the two implicit-GEMM kernels (gemm.hip: 128-row register-staged tiles,
// gemm_glds.hip: 256-row LDS-DMA ring): K-segment row gather and the fused epilogue.
#pragma once
#include "common.h"

namespace vmvg {

constexpr int BK = 64;
// LayerNorm folded into the GEMM with the row statistics taken in its own main loop (vmv.h: VmvGemmParams.ln_eps)
inline bool vmv_gemm_ln_inline(const VmvGemmParams& p) { return !p.rowstat && p.colsum && p.ln_eps > 0.f; }
constexpr int VMV_GLDS_UNSUPPORTED = -100;   // internal: the LDS-DMA kernel cannot address these operands

struct RowInfo {
    int m;       // global row (or -1 when out of range)
    int nb;      // spatial: image base row (n * IH * IW)
    int oy, ox;  // spatial: output pixel
    int fr;      // temporal: frame index
};

VMV_DEV int seg_row_offset(const VmvGemmParams& p, const VmvGemmSeg& sg, const RowInfo& r) {
    // element offset of the source row feeding output row r for this segment, or -1 (zero row)
    if (r.m < 0) return -1;
    if (sg.mode == VMV_SEG_LINEAR) return r.m * sg.ld;
    if (sg.mode == VMV_SEG_SPATIAL) {
        const int iy = r.oy * p.stride + sg.d0;
        const int ix = r.ox * p.stride + sg.d1;
        const int VH = p.IH << p.ups, VW = p.IW << p.ups;
        if (iy < 0 || iy >= VH || ix < 0 || ix >= VW) return -1;
        return (r.nb + (iy >> p.ups) * p.IW + (ix >> p.ups)) * sg.ld;
    }
    // temporal
    const int f = r.fr + sg.d0;
    if (f < 0 || f >= p.F) return -1;
    return (r.m + sg.d0 * p.P) * sg.ld;
}

// Epilogue for 4 consecutive output channels [n, n+4) of row m.  v = accumulators (x half for GEGLU),
// g = gate accumulators (GEGLU only).  `n` indexes W rows (pre-GEGLU numbering).
VMV_DEV void epilogue_store(const VmvGemmParams& p, int m, int n, f32x4_t v, f32x4_t g) {
    if (m >= p.M || n >= p.N) return;
    if (p.rowstat) {       // LayerNorm folded into this GEMM (vmv.h): rstd * (acc - mean * colsum[n])
        const float2 ms = *reinterpret_cast<const float2*>(p.rowstat + (size_t)m * 2);
        const f32x4_t c0 = *reinterpret_cast<const f32x4_t*>(p.colsum + n);
        v = (v - c0 * ms.x) * ms.y;
        if (p.epilogue == VMV_EPI_GEGLU) {
            const f32x4_t c1 = *reinterpret_cast<const f32x4_t*>(p.colsum + n + 16);
            g = (g - c1 * ms.x) * ms.y;
        }
    }
    if (p.bias) {
        const f32x4_t b = *reinterpret_cast<const f32x4_t*>(p.bias + n);
        v += b;
        if (p.epilogue == VMV_EPI_GEGLU) g += *reinterpret_cast<const f32x4_t*>(p.bias + n + 16);
    }
    int no = n;
    if (p.epilogue == VMV_EPI_GEGLU) {
        v.x *= gelu_erf_f(g.x); v.y *= gelu_erf_f(g.y); v.z *= gelu_erf_f(g.z); v.w *= gelu_erf_f(g.w);
        no = (n >> 5) * 16 + (n & 15);
    }
    if (p.rowvec) {
        const f32x4_t rv = *reinterpret_cast<const f32x4_t*>(p.rowvec + (size_t)(m / p.rowvec_div) * p.rowvec_ld + no);
        v += rv;
    }
    act_apply(v, p.act);
    if (p.residual) {
        const u32x2_t r = *reinterpret_cast<const u32x2_t*>(reinterpret_cast<const uint16_t*>(p.residual) + (size_t)m * p.ldr + no);
        const float rs = p.res_scale != 0.f ? p.res_scale : 1.f;
        v.x += rs * elem_lo(r.x); v.y += rs * elem_hi(r.x); v.z += rs * elem_lo(r.y); v.w += rs * elem_hi(r.y);
    }
    if (p.out_fp32) {
        *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldo + no) = v;
    } else {
        u32x2_t o;
        o.x = pack_elem2(v.x, v.y); o.y = pack_elem2(v.z, v.w);
        *reinterpret_cast<u32x2_t*>(reinterpret_cast<uint16_t*>(p.out) + (size_t)m * p.ldo + no) = o;
    }
}


}  // namespace vmvg

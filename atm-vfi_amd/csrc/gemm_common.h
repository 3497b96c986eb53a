// Shared descriptor and epilogue of the two GEMM engines (exact-fp32 MFMA: gemm_conv.hip;
// split-precision f16x3: gemm_f16x3.hip).
#pragma once
#include "common.h"

namespace atmvfi {

struct GemmDev {
    int mode;
    const float* in;
    int in_ld, H, W, Cin;
    long long in_gstride;
    int in_rpg;
    const float* weight;
    int wrows;        // weight rows present (multiple of 16)
    int ktot;         // floats per weight row = taps * cin_pad
    int cin_pad;      // Cin rounded up to 16
    int cpt;          // chunks per tap = cin_pad / 16
    int nchunks;      // taps * cpt
    int Cout;         // real output channels (DECONV: per-position channels)
    int coutp;        // DECONV: Cout rounded up to 4
    int kw, stride, pad, dil;
    int Ho, Wo;
    long long M;
    float* out;
    int out_ld;
    long long out_gstride;
    int out_rpg;
    const int* out_row_map;
    const float* bias;
    const float* prelu;
    const float* in_prelu;
    const float* residual;
    int res_ld;
    // split-precision operands (f16x3 engine only)
    const _Float16* w_hi;
    const _Float16* w_lo;
    int cin_pad32;    // Cin rounded up to 32
    int cpt32;        // 32-channel chunks per tap
    int nchunks32;    // taps * cpt32
    int ktot32;       // halves per split weight row = taps * cin_pad32
};


// Output row of GEMM row m: NHWC pixel (CONV), scattered/grouped token row (LINEAR) or the (0,0)
// position of the 2x2 output patch (DECONV).  Returns false when the row is dropped by the map.
__device__ __forceinline__ bool gemm_out_row(const GemmDev& a, long long m, float*& orow, const float*& rrow) {
    if (a.mode == ATMVFI_GEMM_DECONV) {
        const int hw = a.H * a.W;
        const int n = (int)(m / hw);
        const int rem = (int)(m - (long long)n * hw);
        const int y = rem / a.W;
        const int x = rem - y * a.W;
        orow = a.out + (((long long)n * a.Ho + 2 * y) * a.Wo + 2 * x) * a.out_ld;
    } else {
        long long ro = m;
        if (a.out_row_map) ro = a.out_row_map[m];
        if (ro < 0) return false;
        const long long off = (a.out_rpg > 0) ? (ro / a.out_rpg) * a.out_gstride + (ro % a.out_rpg) * (long long)a.out_ld
                                              : ro * (long long)a.out_ld;
        orow = a.out + off;
    }
    rrow = a.residual ? a.residual + m * (long long)a.res_ld : nullptr;
    return true;
}

// Store four consecutive GEMM columns nb..nb+3 of one row: + bias, PReLU, + residual, 16-byte store.
__device__ __forceinline__ void gemm_store4(const GemmDev& a, float* orow, const float* rrow, int nb, float v0, float v1,
                                            float v2, float v3) {
    int co = nb;
    float* optr = orow;
    if (a.mode == ATMVFI_GEMM_DECONV) {
        if (nb >= 4 * a.coutp) return;
        const int q = (nb >= a.coutp) + (nb >= 2 * a.coutp) + (nb >= 3 * a.coutp);
        co = nb - q * a.coutp;
        optr = orow + ((long long)(q >> 1) * a.Wo + (q & 1)) * a.out_ld;
    }
    if (co >= a.Cout) return;
    const int nvalid = a.Cout - co;   // >= 1
    float vv[4] = {v0, v1, v2, v3};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (e < nvalid) {
            float x = vv[e];
            if (a.bias) x += a.bias[co + e];
            if (a.prelu) x = x > 0.f ? x : a.prelu[co + e] * x;
            if (rrow) x += rrow[co + e];
            vv[e] = x;
        }
    }
    if (nvalid >= 4) {
        *reinterpret_cast<f32x4*>(optr + co) = (f32x4){vv[0], vv[1], vv[2], vv[3]};
    } else {
        for (int e = 0; e < nvalid; ++e) optr[co + e] = vv[e];
    }
}

// gemm_f16x3.hip
int launch_gemm_f16x3(const GemmDev& d, int ngemm, hipStream_t stream);

}  // namespace atmvfi

// 3x3 / stride 1 / pad 1 split-precision convolution, "two workgroups per CU" schedule (gfx950).
//
// Measured on the 512-thread row-stage kernel with s_memtime stamps (tools/stamp_conv.py, N=101 layer):
// the MFMA phase is 43 % of a wave's life; the load-issue burst (12 %), stage barriers (14 %),
// epilogue (12 %), prologue (7 %), halo conversion (6 %) and weight ds_write (5 %) are serialised,
// because one lock-stepped workgroup owns the CU and all 8 waves are always in the same phase.
// This kernel keeps the arithmetic and LDS images (conv3_common.h) but halves the workgroup:
//   * 256 threads = 4 waves, 16x8 output pixels x 16*WN channels; wave w owns rows 2w, 2w+1;
//   * everything single-buffered: halo 23 KB + three taps of weights <= 49 KB -> two workgroups per CU
//     (register budget 256/lane -> 2 waves per SIMD), so one workgroup's load / convert / barrier /
//     epilogue phases run under the other's MFMAs.  Overlap comes from occupancy, not from software
//     pipelining; the only prefetch left is "issue the global loads, then wait at the barrier".
#include "conv3_common.h"

namespace {

constexpr int TH2 = 8, HH2 = TH2 + 2, NPIX2 = HW_ * HH2;       // 18 x 10 = 180 halo pixels
constexpr int HTASKS2 = NPIX2 * 4;
constexpr int HTPT2 = (HTASKS2 + 255) / 256;                   // 3

template <int WN>
__global__ __launch_bounds__(256, 2) void conv3x3_f16x3_half_kernel(const Conv3Dev a) {
    constexpr int BN = 16 * WN;
    constexpr int BP = 3 * BN * 8;                  // 16-byte weight pieces per stage: 3 taps x (hi + lo)
    constexpr int B_PPT = (BP + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
    _Float16* halo_hi = smem;                           // [NPIX2][32]
    _Float16* halo_lo = halo_hi + NPIX2 * 32;
    _Float16* b_hi = halo_lo + NPIX2 * 32;              // [3][BN][32]
    _Float16* b_lo = b_hi + 3 * BN * 32;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 15;
    const int g = lane >> 4;

    int bid = blockIdx.x;
    const int txb = bid % a.tiles_x;
    bid /= a.tiles_x;
    const int tyb = bid % a.tiles_y;
    const int img = bid / a.tiles_y;
    const int ox0 = txb * TW, oy0 = tyb * TH2;
    const int n0 = blockIdx.y * BN;

    // ---- halo tasks ----
    const float* hsrc[HTPT2];
    int hdst[HTPT2], hq[HTPT2];
    bool hok[HTPT2], hact[HTPT2];
#pragma unroll
    for (int k = 0; k < HTPT2; ++k) {
        const int T = tid + 256 * k;
        hact[k] = T < HTASKS2;
        const int hp = hact[k] ? (T >> 2) : 0;
        const int q = T & 3;
        const int hy = hp / HW_, hx = hp - hy * HW_;
        const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
        hok[k] = hact[k] && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        hsrc[k] = a.in + (((long long)img * a.H + (hok[k] ? iy : 0)) * a.W + (hok[k] ? ix : 0)) * a.in_ld + q * 8;
        hdst[k] = hp * 32 + ((q ^ swz64(hp)) << 3);
        hq[k] = q;
    }
    // ---- weight pieces ----
    const _Float16* wsrc[B_PPT];
    int wdst[B_PPT];
    bool wok[B_PPT], wact[B_PPT], wlo[B_PPT];
    const long long ktot = 9ll * a.cin_pad;
#pragma unroll
    for (int k = 0; k < B_PPT; ++k) {
        const int P = tid + 256 * k;
        wact[k] = P < BP;
        const int plane = (P >= 3 * BN * 4) ? 1 : 0;
        int rem = P - plane * 3 * BN * 4;
        const int t = wact[k] ? rem / (BN * 4) : 0;
        rem -= t * BN * 4;
        const int row = wact[k] ? (rem >> 2) : 0;
        const int slot = rem & 3;
        wlo[k] = plane == 1;
        wok[k] = wact[k] && (n0 + row) < a.wrows;
        wsrc[k] = (plane ? a.w_lo : a.w_hi) + (long long)(wok[k] ? n0 + row : 0) * ktot + (long long)t * a.cin_pad + slot * 8;
        wdst[k] = (t * BN + row) * 32 + ((slot ^ swz64(row)) << 3);
    }

    f32x4 acc[2][WN], cor[2][WN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            cor[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }

    const int nchunks = a.cin_pad >> 5;

    for (int chunk = 0; chunk < nchunks; ++chunk) {
#pragma unroll 1
        for (int ky = 0; ky < 3; ++ky) {
            // ---- load phase: issue every global load, then meet the other waves, then fill LDS ----
            f16x8 wr[B_PPT];
            const long long koff = (long long)(ky * 3) * a.cin_pad + chunk * 32;
#pragma unroll
            for (int k = 0; k < B_PPT; ++k) wr[k] = *reinterpret_cast<const f16x8*>(wsrc[k] + koff);   // row-clamped: always valid
            f32x4 hr[HTPT2][2];
            int hnv[HTPT2];
            if (ky == 0) {
#pragma unroll
                for (int k = 0; k < HTPT2; ++k) {
                    const int c = chunk * 32 + hq[k] * 8;
                    const bool ok = hok[k] && c < a.Cin;
                    const int nv = ok ? a.Cin - c : 0;
                    const float* p = ok ? hsrc[k] + chunk * 32 : a.in;      // masked lanes read the tensor base
                    hr[k][0] = *reinterpret_cast<const f32x4*>(p);
                    hr[k][1] = *reinterpret_cast<const f32x4*>(p + (nv > 4 ? 4 : 0));
                    hnv[k] = nv;
                }
            }
            __syncthreads();        // previous stage's fragment reads are done: LDS may be overwritten
#pragma unroll
            for (int k = 0; k < B_PPT; ++k)
                if (wact[k])
                    *reinterpret_cast<f16x8*>((wlo[k] ? b_lo : b_hi) + wdst[k]) = wok[k] ? wr[k] : (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
            if (ky == 0) {
#pragma unroll
                for (int k = 0; k < HTPT2; ++k) {
                    if (hact[k]) {
                        f32x4 va = hr[k][0], vb = hr[k][1];
                        const int nv = hnv[k];
                        va.x = nv > 0 ? va.x : 0.f;
                        va.y = nv > 1 ? va.y : 0.f;
                        va.z = nv > 2 ? va.z : 0.f;
                        va.w = nv > 3 ? va.w : 0.f;
                        vb.x = nv > 4 ? vb.x : 0.f;
                        vb.y = nv > 5 ? vb.y : 0.f;
                        vb.z = nv > 6 ? vb.z : 0.f;
                        vb.w = nv > 7 ? vb.w : 0.f;
                        f16x8 hi, lo;
                        split8(va, vb, hi, lo);
                        *reinterpret_cast<f16x8*>(halo_hi + hdst[k]) = hi;
                        *reinterpret_cast<f16x8*>(halo_lo + hdst[k]) = lo;
                    }
                }
            }
            __syncthreads();

            // ---- compute phase: the three taps of kernel row ky ----
            f16x8 xhA[2], xlA[2], xhB[2], xlB[2];
            auto load_x = [&](int t, f16x8 (&h)[2], f16x8 (&l)[2]) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int p = (2 * wave + i + ky) * HW_ + t + r;
                    const int off = p * 32 + ((g ^ swz64(p)) << 3);
                    h[i] = *reinterpret_cast<const f16x8*>(halo_hi + off);
                    l[i] = *reinterpret_cast<const f16x8*>(halo_lo + off);
                }
            };
            auto tap = [&](int t, const f16x8 (&xh)[2], const f16x8 (&xl)[2]) {
                const int wbase = t * BN * 32 + r * 32 + ((g ^ swz64(r)) << 3);
                f16x8 wh[2], wl[2];
                wh[0] = *reinterpret_cast<const f16x8*>(b_hi + wbase);
                wl[0] = *reinterpret_cast<const f16x8*>(b_lo + wbase);
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    if (j + 1 < WN) {
                        wh[(j + 1) & 1] = *reinterpret_cast<const f16x8*>(b_hi + wbase + (j + 1) * 16 * 32);
                        wl[(j + 1) & 1] = *reinterpret_cast<const f16x8*>(b_lo + wbase + (j + 1) * 16 * 32);
                    }
                    const f16x8 ch = wh[j & 1], cl = wl[j & 1];
                    cor[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cl, xh[0], cor[0][j], 0, 0, 0);
                    cor[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cl, xh[1], cor[1][j], 0, 0, 0);
                    acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch, xh[0], acc[0][j], 0, 0, 0);
                    acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch, xh[1], acc[1][j], 0, 0, 0);
                    cor[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch, xl[0], cor[0][j], 0, 0, 0);
                    cor[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch, xl[1], cor[1][j], 0, 0, 0);
                }
            };
            load_x(0, xhA, xlA);
            load_x(1, xhB, xlB);
            tap(0, xhA, xlA);
            load_x(2, xhA, xlA);
            tap(1, xhB, xlB);
            tap(2, xhA, xlA);
        }
    }

    // ---- epilogue ----
    float* orow[2];
    bool live[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int oy = oy0 + 2 * wave + i, ox = ox0 + r;
        live[i] = oy < a.H && ox < a.W;
        orow[i] = a.out + (((long long)img * a.H + (live[i] ? oy : 0)) * a.W + (live[i] ? ox : 0)) * a.out_ld;
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int co = n0 + 16 * j + 4 * g;
        const int nvalid = a.Cout - co;
        if (nvalid <= 0) continue;
        f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f}, pv = (f32x4){1.f, 1.f, 1.f, 1.f};
        if (nvalid >= 4) {
            if (a.bias) bv = *reinterpret_cast<const f32x4*>(a.bias + co);
            if (a.prelu) pv = *reinterpret_cast<const f32x4*>(a.prelu + co);
        } else {
            float bb[4] = {0.f, 0.f, 0.f, 0.f}, pp[4] = {1.f, 1.f, 1.f, 1.f};
            for (int e = 0; e < nvalid; ++e) {
                if (a.bias) bb[e] = a.bias[co + e];
                if (a.prelu) pp[e] = a.prelu[co + e];
            }
            bv = (f32x4){bb[0], bb[1], bb[2], bb[3]};
            pv = (f32x4){pp[0], pp[1], pp[2], pp[3]};
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (!live[i]) continue;
            f32x4 v = acc[i][j] + cor[i][j] * LO_UNSCALE + bv;
            v.x = v.x > 0.f ? v.x : pv.x * v.x;
            v.y = v.y > 0.f ? v.y : pv.y * v.y;
            v.z = v.z > 0.f ? v.z : pv.z * v.z;
            v.w = v.w > 0.f ? v.w : pv.w * v.w;
            if (nvalid >= 4) {
                *reinterpret_cast<f32x4*>(orow[i] + co) = v;
            } else {
                const float vv[4] = {v.x, v.y, v.z, v.w};
                for (int e = 0; e < nvalid; ++e) orow[i][co + e] = vv[e];
            }
        }
    }
}

template <int WN>
int launch_half(const Conv3Dev& d0, int ntiles, hipStream_t s) {
    constexpr int BN = 16 * WN;
    Conv3Dev d = d0;
    d.tiles_y = (d.H + TH2 - 1) / TH2;
    const size_t lds = (size_t)(2 * NPIX2 * 32 + 2 * 3 * BN * 32) * sizeof(_Float16);
    auto kern = conv3x3_f16x3_half_kernel<WN>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        ATMVFI_REQUIRE(e == hipSuccess, ATMVFI_ELAUNCH, "conv3x3_f16x3_half: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    const long long gx = (long long)d.N * d.tiles_x * d.tiles_y;
    ATMVFI_REQUIRE(gx < (1ll << 31), ATMVFI_EINVAL, "conv3x3_f16x3_half: grid too large");
    dim3 grid((unsigned)gx, (unsigned)((ntiles + WN - 1) / WN));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, d);
    return atmvfi::check_launch("conv3x3_f16x3_half");
}

}  // namespace

int atmvfi::launch_conv3x3_half(const Conv3Dev& d, int ntiles, hipStream_t s) {
    int best = 1;
    float best_cost = 1e30f;
    for (int wn = 1; wn <= 8; ++wn) {
        const int padded = (ntiles + wn - 1) / wn * wn;
        const float cost = (float)padded * (1.0f + 1.0f / (float)wn);
        if (cost <= best_cost) { best_cost = cost; best = wn; }
    }
    switch (best) {
        case 1: return launch_half<1>(d, ntiles, s);
        case 2: return launch_half<2>(d, ntiles, s);
        case 3: return launch_half<3>(d, ntiles, s);
        case 4: return launch_half<4>(d, ntiles, s);
        case 5: return launch_half<5>(d, ntiles, s);
        case 6: return launch_half<6>(d, ntiles, s);
        case 7: return launch_half<7>(d, ntiles, s);
        default: return launch_half<8>(d, ntiles, s);
    }
}

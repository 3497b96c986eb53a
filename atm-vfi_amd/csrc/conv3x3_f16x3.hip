// 3x3 / stride 1 / pad 1 convolution for gfx950 with an LDS-resident input halo and
// split-precision MFMA ("f16x3").
//
// Why: the generic implicit-GEMM kernel re-reads every input pixel once per tap (9x) from
// L2 and runs on the exact-fp32 MFMA, which is 1/16 of the 16-bit MFMA rate.  3x3 s1 convs
// are two thirds of the network's FLOPs, so this kernel
//   * stages the (16+2)x(16+2) input halo of a 16x16 output tile ONCE per 32-channel chunk
//     and walks the 9 taps by shifting the fragment base address inside LDS;
//   * splits every fp32 operand as x = hi + lo'/1024 (hi = fp16(x), lo' = fp16((x - hi)*1024)) while
//     staging, and accumulates hi*hi into one fp32 accumulator and hi*lo' + lo'*hi into a second
//     one (folded in with 2^-10 in the epilogue) on v_mfma_f32_16x16x32_f16: 3 MFMAs at 16x the
//     fp32-MFMA rate = 5.3x, ~22 significand bits at any operand magnitude (the dropped lo*lo
//     term is 2^-22 relative).  Weights are split once at pack time.
//
// Block = 512 threads = 8 wavefronts; output tile = 16x16 pixels (wave w owns output rows
// 2w, 2w+1 = two 16-pixel MFMA column tiles) x 16*WN output channels.
// LDS (one __shared__ array): halo hi/lo planes [2 buffers][324 px][32 halves] and weight
// hi/lo planes [2 buffers][16*WN rows][32 halves]; rows are 64 bytes and 16-byte slots are
// XOR-swizzled with ((row>>2)&1)<<1, which keeps ds_read_b128 fragment reads conflict-free
// for EVERY base row (the tap shift moves the base).  One barrier per (chunk, tap) stage;
// the next stage's weights and one third of the next chunk's halo are fetched into
// registers before the MFMA phase and written to the other LDS buffer after it.
#include "common.h"

#include "conv3_common.h"

#include <stdlib.h>

namespace {

template <int WN>
__global__ __launch_bounds__(512) void conv3x3_f16x3_kernel(const Conv3Dev a) {
    constexpr int BN = 16 * WN;
    constexpr int BP = BN * 8;                      // 16-byte weight pieces per stage (hi + lo)
    constexpr int B_PPT = (BP + 511) / 512;
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
    _Float16* halo_hi = smem;                           // [2][NPIX][32]
    _Float16* halo_lo = halo_hi + 2 * NPIX * 32;
    _Float16* b_hi = halo_lo + 2 * NPIX * 32;           // [2][BN][32]
    _Float16* b_lo = b_hi + 2 * BN * 32;

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
    const int ox0 = txb * TW, oy0 = tyb * TH;
    const int n0 = blockIdx.y * BN;

    // ---- halo task bookkeeping: task T = tid + 512*k -> (halo pixel, 8-channel group) ----
    const float* hsrc[HALO_TPT];
    int hdst[HALO_TPT];       // halves offset inside one halo plane buffer
    bool hok[HALO_TPT], hact[HALO_TPT];
    int hq[HALO_TPT];
#pragma unroll
    for (int k = 0; k < HALO_TPT; ++k) {
        const int T = tid + 512 * k;
        hact[k] = T < HALO_TASKS;
        const int hp = hact[k] ? (T >> 2) : 0;
        const int q = T & 3;
        const int hy = hp / HW_, hx = hp - hy * HW_;
        const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
        hok[k] = hact[k] && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        hsrc[k] = a.in + (((long long)img * a.H + (hok[k] ? iy : 0)) * a.W + (hok[k] ? ix : 0)) * a.in_ld + q * 8;
        hdst[k] = hp * 32 + ((q ^ swz64(hp)) << 3);
        hq[k] = q;
    }
    // ---- weight piece bookkeeping: piece P = tid + 512*k -> (plane, row, slot) ----
    const _Float16* wsrc[B_PPT];
    int wdst[B_PPT];
    bool wok[B_PPT], wact[B_PPT], wlo[B_PPT];
    const long long ktot = 9ll * a.cin_pad;
#pragma unroll
    for (int k = 0; k < B_PPT; ++k) {
        const int P = tid + 512 * k;
        wact[k] = P < BP;
        const int plane = (P >= BN * 4) ? 1 : 0;
        const int rem = P - plane * BN * 4;
        const int row = wact[k] ? (rem >> 2) : 0;
        const int slot = rem & 3;
        wlo[k] = plane == 1;
        wok[k] = wact[k] && (n0 + row) < a.wrows;
        wsrc[k] = (plane ? a.w_lo : a.w_hi) + (long long)(wok[k] ? n0 + row : 0) * ktot + slot * 8;
        wdst[k] = row * 32 + ((slot ^ swz64(row)) << 3);
    }

    f32x4 acc[2][WN], cor[2][WN];      // hi*hi terms / (hi*lo' + lo'*hi) terms, the latter scaled by 1024
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            cor[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }

    const int nchunks = a.cin_pad >> 5;
    const int nstages = nchunks * 9;

    // Register-staged prefetch, deep enough to cover loaded-memory latency with one block per CU:
    // weights run TWO stages ahead (two register sets, alternating), the next chunk's halo is
    // requested at tap 0 and converted/written at tap HALO_WRITE_TAP.
    constexpr int HALO_WRITE_TAP = 5;
    f32x4 hr[HALO_TPT][2];
    int hnv[HALO_TPT];
    f16x8 wrA[B_PPT], wrB[B_PPT];

    auto halo_load = [&](int k, int chunk) {        // k is a compile-time constant at every call site
        // Unconditional loads from a clamped, always-valid address, zeroed afterwards by selects: a
        // predicated load makes hipcc branch around it and wait vmcnt(0) on the spot, which serialises
        // the whole prefetch (seen in the ISA; cdna_hip_programming.md "Three .s-level traps" (c)).
        const int c = chunk * 32 + hq[k] * 8;
        const bool ok = hok[k] && c < a.Cin;
        const int nv = ok ? a.Cin - c : 0;                       // valid channels in this group of 8
        const float* p = ok ? hsrc[k] + chunk * 32 : a.in;       // masked lanes read the tensor base: hsrc[k] + q*8 may lie past a narrow last pixel
        const f32x4 va = *reinterpret_cast<const f32x4*>(p);
        const f32x4 vb = *reinterpret_cast<const f32x4*>(p + (nv > 4 ? 4 : 0));
        hnv[k] = nv;                                             // masking happens at store time: no early consumer
        hr[k][0] = va;
        hr[k][1] = vb;
    };
    auto halo_store = [&](int k, int buf) {
        if (hact[k]) {
            f16x8 hi, lo;
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
            split8(va, vb, hi, lo);
            *reinterpret_cast<f16x8*>(halo_hi + buf * NPIX * 32 + hdst[k]) = hi;
            *reinterpret_cast<f16x8*>(halo_lo + buf * NPIX * 32 + hdst[k]) = lo;
        }
    };
    auto w_load = [&](int stage, f16x8 (&dst)[B_PPT]) {
        const int chunk = stage / 9;
        const int tap = stage - chunk * 9;
        const long long koff = (long long)tap * a.cin_pad + chunk * 32;
#pragma unroll
        for (int k = 0; k < B_PPT; ++k) {
            dst[k] = *reinterpret_cast<const f16x8*>(wsrc[k] + koff);        // row-clamped address: always valid; masked at store
        }
    };
    auto w_store = [&](int buf, const f16x8 (&src)[B_PPT]) {
#pragma unroll
        for (int k = 0; k < B_PPT; ++k)
            if (wact[k])
                *reinterpret_cast<f16x8*>((wlo[k] ? b_lo : b_hi) + buf * BN * 32 + wdst[k]) =
                    wok[k] ? src[k] : (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
    };
    // one (chunk, tap) stage: `nxt` holds the weights of stage s+1 (written to LDS after the MFMAs),
    // `far` receives the weights of stage s+2.
    auto stage = [&](int s, f16x8 (&nxt)[B_PPT], f16x8 (&far)[B_PPT]) {
        const int chunk = s / 9;
        const int tap = s - chunk * 9;
        const int ky = tap / 3, kx = tap - ky * 3;
        const int hb = chunk & 1, wb = s & 1;
        const bool more_h = chunk + 1 < nchunks;
        if (s + 2 < nstages) w_load(s + 2, far);
        if (more_h && tap == 0) {
#pragma unroll
            for (int k = 0; k < HALO_TPT; ++k) halo_load(k, chunk + 1);
        }
        f16x8 xh[2], xl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = (2 * wave + i + ky) * HW_ + kx + r;
            const int off = hb * NPIX * 32 + p * 32 + ((g ^ swz64(p)) << 3);
            xh[i] = *reinterpret_cast<const f16x8*>(halo_hi + off);
            xl[i] = *reinterpret_cast<const f16x8*>(halo_lo + off);
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int row = 16 * j + r;
            const int off = wb * BN * 32 + row * 32 + ((g ^ swz64(row)) << 3);
            const f16x8 wh = *reinterpret_cast<const f16x8*>(b_hi + off);
            const f16x8 wl = *reinterpret_cast<const f16x8*>(b_lo + off);
                // dependent MFMAs (same accumulator) are kept 4 issues apart: back-to-back they stall the pipe
                cor[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[0], cor[0][j], 0, 0, 0);
                cor[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[1], cor[1][j], 0, 0, 0);
                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[0], acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[1], acc[1][j], 0, 0, 0);
                cor[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[0], cor[0][j], 0, 0, 0);
                cor[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[1], cor[1][j], 0, 0, 0);
        }
        if (s + 1 < nstages) w_store(wb ^ 1, nxt);
        if (more_h && tap == HALO_WRITE_TAP) {
#pragma unroll
            for (int k = 0; k < HALO_TPT; ++k) halo_store(k, hb ^ 1);
        }
        __syncthreads();
    };

    // ---- prologue: halo of chunk 0, weights of stage 0 in LDS, weights of stage 1 in flight ----
#pragma unroll
    for (int k = 0; k < HALO_TPT; ++k) halo_load(k, 0);
    w_load(0, wrA);
    if (nstages > 1) w_load(1, wrB);
#pragma unroll
    for (int k = 0; k < HALO_TPT; ++k) halo_store(k, 0);
    w_store(0, wrA);
    __syncthreads();

    for (int s = 0; s < nstages; s += 2) {
        stage(s, wrB, wrA);                        // even stage: stage s+1 lives in wrB, s+2 goes to wrA
        if (s + 1 < nstages) stage(s + 1, wrA, wrB);
    }

    // ---- epilogue: lane holds channels nb..nb+3 of pixel (oy0 + 2*wave + i, ox0 + r) ----
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
        // bias / PReLU slopes of the tile, one 16-byte load each (host guarantees alignment)
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

__global__ void pack_split_kernel(int mode, const float* __restrict__ src, _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                  int Cout, int Cin, int kh, int kw, int rows, int cin_pad, int coutp) {
    const int taps = (mode == ATMVFI_GEMM_DECONV) ? 1 : kh * kw;
    const long long total = (long long)rows * taps * cin_pad;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % cin_pad);
        const int tap = (int)((idx / cin_pad) % taps);
        const int row = (int)(idx / ((long long)cin_pad * taps));
        float v = 0.f;
        if (c < Cin) {
            if (mode == ATMVFI_GEMM_DECONV) {
                const int q = row / coutp;
                const int co = row - q * coutp;
                if (q < 4 && co < Cout) v = src[(((long long)c * Cout + co) * 2 + (q >> 1)) * 2 + (q & 1)];   // IOHW
            } else if (row < Cout) {
                const int ky = tap / kw, kx = tap - ky * kw;
                v = src[(((long long)row * Cin + c) * kh + ky) * kw + kx];                                      // OIHW
            }
        }
        const _Float16 h = sat_half(v);
        hi[idx] = h;
        lo[idx] = sat_half((v - (float)h) * LO_SCALE);
    }
}

template <int WN>
int launch3(const Conv3Dev& d, int ntiles, hipStream_t s) {
    constexpr int BN = 16 * WN;
    const size_t lds = (size_t)(4 * NPIX * 32 + 4 * BN * 32) * sizeof(_Float16);
    auto kern = conv3x3_f16x3_kernel<WN>;
    static bool attr_set = false;      // idempotent, per instantiation
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        ATMVFI_REQUIRE(e == hipSuccess, ATMVFI_ELAUNCH, "conv3x3_f16x3: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    dim3 grid((unsigned)((long long)d.N * d.tiles_x * d.tiles_y), (unsigned)((ntiles + WN - 1) / WN));
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, d);
    return atmvfi::check_launch("conv3x3_f16x3");
}

}  // namespace

static int split_rows(int mode, int Cout) {
    return (mode == ATMVFI_GEMM_DECONV) ? atmvfi::round_up(4 * atmvfi::round_up(Cout, 4), 16) : atmvfi::round_up(Cout, 16);
}

extern "C" int64_t atmvfi_split_weight_halves(int mode, int Cout, int Cin, int kh, int kw) {
    const int taps = (mode == ATMVFI_GEMM_DECONV) ? 1 : kh * kw;
    return (int64_t)split_rows(mode, Cout) * taps * atmvfi::round_up(Cin, 32);
}

extern "C" int atmvfi_pack_weight_split(int mode, const float* src, void* dst_hi, void* dst_lo, int Cout, int Cin, int kh,
                                         int kw, void* stream) {
    ATMVFI_REQUIRE(src && dst_hi && dst_lo && Cout > 0 && Cin > 0 && kh > 0 && kw > 0 && mode >= 0 && mode <= 2, ATMVFI_EINVAL,
                   "pack_weight_split: bad arguments");
    if (mode == ATMVFI_GEMM_DECONV) ATMVFI_REQUIRE(kh == 2 && kw == 2, ATMVFI_EINVAL, "pack_weight_split: deconv must be 2x2");
    const int64_t total = atmvfi_split_weight_halves(mode, Cout, Cin, kh, kw);
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_split_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mode, src, (_Float16*)dst_hi,
                       (_Float16*)dst_lo, Cout, Cin, kh, kw, split_rows(mode, Cout), atmvfi::round_up(Cin, 32),
                       atmvfi::round_up(Cout, 4));
    return atmvfi::check_launch("pack_weight_split");
}

extern "C" int atmvfi_conv3x3_f16x3(const float* in, int in_ld, int N, int H, int W, int Cin, const void* w_hi,
                                     const void* w_lo, int Cout, float* out, int out_ld, const float* bias,
                                     const float* prelu, void* stream) {
    ATMVFI_REQUIRE(in && w_hi && w_lo && out, ATMVFI_EINVAL, "conv3x3_f16x3: null pointer");
    ATMVFI_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, ATMVFI_EINVAL, "conv3x3_f16x3: bad shape");
    ATMVFI_REQUIRE(in_ld % 4 == 0 && in_ld >= atmvfi::round_up(Cin, 4) && out_ld % 4 == 0 && out_ld >= atmvfi::round_up(Cout, 4),
                   ATMVFI_EALIGN, "conv3x3_f16x3: leading dimensions must be multiples of 4 and cover the channels");
    ATMVFI_REQUIRE(atmvfi::aligned16(in) && atmvfi::aligned16(out) && atmvfi::aligned16(w_hi) && atmvfi::aligned16(w_lo) &&
                       (!bias || atmvfi::aligned16(bias)) && (!prelu || atmvfi::aligned16(prelu)),
                   ATMVFI_EALIGN, "conv3x3_f16x3: pointers (incl. bias/prelu) must be 16-byte aligned");
    Conv3Dev d;
    d.in = in; d.in_ld = in_ld; d.N = N; d.H = H; d.W = W; d.Cin = Cin;
    d.w_hi = (const _Float16*)w_hi; d.w_lo = (const _Float16*)w_lo;
    d.wrows = atmvfi::round_up(Cout, 16);
    d.cin_pad = atmvfi::round_up(Cin, 32);
    d.Cout = Cout; d.out = out; d.out_ld = out_ld; d.bias = bias; d.prelu = prelu;
    d.stamp = nullptr;
    d.tiles_x = (W + TW - 1) / TW;
    d.tiles_y = (H + TH - 1) / TH;
    ATMVFI_REQUIRE((long long)N * d.tiles_x * d.tiles_y < (1ll << 31), ATMVFI_EINVAL, "conv3x3_f16x3: grid too large");
    const int ntiles = (Cout + 15) / 16;
    hipStream_t s = (hipStream_t)stream;
    // schedules (A/B via ATMVFI_CONV3_SCHED): "row" (default) = one 512-thread workgroup per CU, 3 taps per stage;
    // "half" = two 256-thread workgroups per CU on 16x8 tiles, single-buffered (wins only around WN = 6, spills at
    // WN >= 7); "onetap" = the first schedule, one tap per stage
    static const char* sched_env = getenv("ATMVFI_CONV3_SCHED");
    static const int sched = !sched_env ? 1 : (sched_env[0] == 'h' ? 0 : (sched_env[0] == 'o' ? 2 : 1));
    if (sched == 0) return atmvfi::launch_conv3x3_half(d, ntiles, s);
    if (sched == 1) return atmvfi::launch_conv3x3_row(d, ntiles, s);
    int best = 1;
    float best_cost = 1e30f;
    for (int wn = 1; wn <= 8; ++wn) {
        const int padded = (ntiles + wn - 1) / wn * wn;
        const float cost = (float)padded * (1.0f + 1.0f / (float)wn);
        if (cost <= best_cost) { best_cost = cost; best = wn; }
    }
    switch (best) {
        case 1: return launch3<1>(d, ntiles, s);
        case 2: return launch3<2>(d, ntiles, s);
        case 3: return launch3<3>(d, ntiles, s);
        case 4: return launch3<4>(d, ntiles, s);
        case 5: return launch3<5>(d, ntiles, s);
        case 6: return launch3<6>(d, ntiles, s);
        case 7: return launch3<7>(d, ntiles, s);
        default: return launch3<8>(d, ntiles, s);
    }
}

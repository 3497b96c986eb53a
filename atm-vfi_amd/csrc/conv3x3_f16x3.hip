// 3x3 / stride 1 / pad 1 convolution for gfx950 with an LDS-resident input halo and split-precision MFMA ("f16x3"):
// host entry point, weight layouts and pack kernels.  The kernel itself is conv3x3_f16x3_row.hip.
//
// Why a dedicated kernel: the generic implicit-GEMM engine re-reads every input pixel once per tap (9x) from L2; 3x3 s1
// convolutions are two thirds of the network's FLOPs, so this one
//   * stages the (16+2)x(16+2) input halo of a 16x16 output tile ONCE per 32-channel chunk and walks the 9 taps by
//     shifting the fragment base address inside LDS;
//   * splits every fp32 operand as x = hi + lo'/1024 (hi = fp16(x), lo' = fp16((x - hi)*1024)) while staging, and
//     accumulates hi*hi into one fp32 accumulator and hi*lo' + lo'*hi into a second one (folded in with 2^-10 in the
//     epilogue) on v_mfma_f32_16x16x32_f16: 3 MFMAs at 16x the fp32-MFMA rate, ~22 significand bits at any operand
//     magnitude (the dropped lo*lo term is 2^-22 relative).  Weights are split once at pack time.
//
// Weight layout (atmvfi_pack_weight_conv3x3), fp16 hi and lo' planes, K-STEP MAJOR: [k-step][row (padded to 16)][32 halves].
//   CF = 32*floor(Cin/32) and tail = Cin - CF if 1 <= tail <= 8, else CF = round_up(Cin, 32), tail = 0.
//   k-step (chunk c < CF/32, tap q < 9) has index 9c + q and holds w[row][32c .. 32c+31][tap q]; with a tail, three more k-steps
//   9*CF/32 + t (t = 0..2) hold the tap-packed tail: entry [4 taps 4t..4t+3][8 channels CF..CF+7] (taps >= 9, channels >= Cin: 0).
//   The 16 rows x 64 bytes one LDS-DMA instruction moves are therefore ONE contiguous KiB (8 full cache lines); in a row-major
//   layout they were 16 half-used lines, and issuing those was a third of the half-tile schedule's time (tools/stamp_conv.py).
// The decoder widths are 32k+5 (101, 197, 389: features + two flows + mask), so a plain 32-channel last chunk would
// spend 9 k-steps (one per tap) on 5 live channels; the tail steps spend 3 (k = 4 taps x 8 channels each), which removes
// 6 of 36 / 63 / 117 k-steps of those layers.
#include "common.h"

#include "conv3_common.h"

#include <stdlib.h>

namespace {

// generic split planes, K-STEP MAJOR: [k-step = tap * (CinPad32 / 32) + chunk][row][32 halves] -- the 16 rows x 64 B that one
// load / LDS-DMA instruction of the GEMM engines moves are one contiguous KiB (gemm_f16x3.hip, gemm_split.hip)
__global__ void pack_split_kernel(int mode, const float* __restrict__ src, _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                  int Cout, int Cin, int kh, int kw, int rows, int cin_pad, int coutp) {
    fp16_saturate_on();
    const int taps = (mode == ATMVFI_GEMM_DECONV) ? 1 : kh * kw;
    const int cpt = cin_pad >> 5;
    const long long total = (long long)rows * taps * cin_pad;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 31);
        const int row = (int)((idx >> 5) % rows);
        const int kstep = (int)((idx >> 5) / rows);
        const int tap = kstep / cpt;
        const int c = (kstep - tap * cpt) * 32 + e;
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

// conv3x3 layout: k-step major, tap-packed tail (see the file header)
__global__ void pack_conv3x3_kernel(const float* __restrict__ src, _Float16* __restrict__ hi, _Float16* __restrict__ lo, int Cout, int Cin,
                                    int rows, int cf, int tail) {
    fp16_saturate_on();
    const int nsteps = 9 * (cf >> 5) + (tail ? 3 : 0);
    const long long total = (long long)nsteps * rows * 32;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 31);
        const int row = (int)((idx >> 5) % rows);
        const int step = (int)((idx >> 5) / rows);
        int tap, c;
        if (step < 9 * (cf >> 5)) {
            tap = step % 9;
            c = (step / 9) * 32 + e;
        } else {                                // tail step t: [4 taps][8 channels]
            const int t = step - 9 * (cf >> 5);
            tap = 4 * t + (e >> 3);
            c = (e & 7) < tail ? cf + (e & 7) : Cin;
        }
        float v = 0.f;
        if (row < Cout && tap < 9 && c < Cin) v = src[((long long)row * Cin + c) * 9 + tap];     // OIHW, tap = ky*3 + kx
        const _Float16 h = sat_half(v);
        hi[idx] = h;
        lo[idx] = sat_half((v - (float)h) * LO_SCALE);
    }
}

}  // namespace

// full 32-channel chunks and tail channels of the conv3x3 layout
static void conv3_layout(int Cin, int& cf, int& tail) {
    const int t = Cin % 32;
    if (t >= 1 && t <= 8) { cf = Cin - t; tail = t; }
    else { cf = atmvfi::round_up(Cin, 32); tail = 0; }
}
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
                                     const float* prelu, void* out_hi, void* out_lo, int64_t plane_rows,
                                     const float* plane_prelu, int schedule, int wn, void* stream) {
    ATMVFI_REQUIRE(in && w_hi && w_lo && out, ATMVFI_EINVAL, "conv3x3_f16x3: null pointer");
    ATMVFI_REQUIRE(schedule >= -1 && schedule <= 1 && wn >= 0 && wn <= 8, ATMVFI_EINVAL,
                   "conv3x3_f16x3: schedule -1 (auto), 0 (row) or 1 (half), wn 0 (auto) or 1..8; got %d, %d", schedule, wn);
    ATMVFI_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, ATMVFI_EINVAL, "conv3x3_f16x3: bad shape");
    ATMVFI_REQUIRE(in_ld % 4 == 0 && in_ld >= atmvfi::round_up(Cin, 4) && out_ld % 4 == 0 && out_ld >= atmvfi::round_up(Cout, 4),
                   ATMVFI_EALIGN, "conv3x3_f16x3: leading dimensions must be multiples of 4 and cover the channels");
    ATMVFI_REQUIRE(atmvfi::aligned16(in) && atmvfi::aligned16(out) && atmvfi::aligned16(w_hi) && atmvfi::aligned16(w_lo) &&
                       (!bias || atmvfi::aligned16(bias)) && (!prelu || atmvfi::aligned16(prelu)),
                   ATMVFI_EALIGN, "conv3x3_f16x3: pointers (incl. bias/prelu) must be 16-byte aligned");
    ATMVFI_REQUIRE((out_hi == nullptr) == (out_lo == nullptr), ATMVFI_EINVAL, "conv3x3_f16x3: plane sink needs both planes");
    if (out_hi) {
        ATMVFI_REQUIRE(plane_rows >= (int64_t)N * H * W, ATMVFI_EINVAL, "conv3x3_f16x3: plane_rows %lld < N*H*W", (long long)plane_rows);
        ATMVFI_REQUIRE(atmvfi::aligned16(out_hi) && atmvfi::aligned16(out_lo) && (!plane_prelu || atmvfi::aligned16(plane_prelu)),
                       ATMVFI_EALIGN, "conv3x3_f16x3: plane sink pointers must be 16-byte aligned");
    }
    Conv3Dev d;
    d.in = in; d.in_ld = in_ld; d.N = N; d.H = H; d.W = W; d.Cin = Cin;
    d.w_hi = (const _Float16*)w_hi; d.w_lo = (const _Float16*)w_lo;
    d.wrows = atmvfi::round_up(Cout, 16);
    conv3_layout(Cin, d.cf, d.tail);
    d.cs = 0;
    d.ktot = 0;
    d.Cout = Cout; d.out = out; d.out_ld = out_ld; d.bias = bias; d.prelu = prelu;
    d.out_hi = (_Float16*)out_hi; d.out_lo = (_Float16*)out_lo; d.plane_rows = plane_rows; d.plane_prelu = plane_prelu;
    d.stamp = nullptr;
    d.nblocks = 0;
    d.tchunk = 0;
    d.legacy_order = 0;
    d.force_schedule = schedule;
    d.force_wn = wn;
    d.tiles_x = (W + TW - 1) / TW;
    d.tiles_y = 0;                                  // set by the launcher: the tile height depends on the schedule
    ATMVFI_REQUIRE((long long)N * d.tiles_x * ((H + 7) / 8) * 8 < (1ll << 31), ATMVFI_EINVAL, "conv3x3_f16x3: grid too large");
    return atmvfi::launch_conv3x3_row(d, (Cout + 15) / 16, (hipStream_t)stream);
}

extern "C" int64_t atmvfi_conv3x3_weight_halves(int Cout, int Cin) {
    int cf, tail;
    conv3_layout(Cin, cf, tail);
    return (int64_t)atmvfi::round_up(Cout, 16) * 32 * (9 * (cf >> 5) + (tail ? 3 : 0));
}

extern "C" int atmvfi_pack_weight_conv3x3(const float* src, void* dst_hi, void* dst_lo, int Cout, int Cin, void* stream) {
    ATMVFI_REQUIRE(src && dst_hi && dst_lo && Cout > 0 && Cin > 0, ATMVFI_EINVAL, "pack_weight_conv3x3: bad arguments");
    int cf, tail;
    conv3_layout(Cin, cf, tail);
    const int64_t total = atmvfi_conv3x3_weight_halves(Cout, Cin);
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_conv3x3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (_Float16*)dst_hi, (_Float16*)dst_lo,
                       Cout, Cin, atmvfi::round_up(Cout, 16), cf, tail);
    return atmvfi::check_launch("pack_weight_conv3x3");
}

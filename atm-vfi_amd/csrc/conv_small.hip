// Direct 3x3 / stride 1 / pad 1 convolution for a handful of input channels (the network's first layer: 3 -> 24 / 16 on the
// NHWC4-packed frames, network_base.py:20-25 conv() via feat_extracts.0.0).  As an implicit GEMM this layer is nine k-steps of
// three live channels each on the fp32 MFMA (0.48 ms at 1080p, 11 TF/s); it is really a store-bound layer (33 MB in, 400 MB out),
// so it runs on the vector ALU instead: one thread per output pixel, its 9*Cin inputs in registers, the weights broadcast from LDS
// (every lane reads the same float4), exact fp32 multiply-add in (tap, channel) order per output, bias + PReLU fused.
#include "common.h"
#include "gemm_common.h"

namespace {

using atmvfi::GemmDev;

// weights: the packed fp32 GEMM layout [rows16][9][cin_pad16] (atmvfi_pack_weight)
template <int CIN, int CO4>      // CO4 = output channels / 4
__global__ __launch_bounds__(256) void conv3x3_small_kernel(const GemmDev a) {
    constexpr int COUT = 4 * CO4;
    constexpr int K = 9 * CIN;
    __shared__ __attribute__((aligned(16))) float wl[K * COUT];           // [k = tap*CIN + c][cout]
    __shared__ __attribute__((aligned(16))) float tile[18 * 18 * 4];      // halo, 4 floats per pixel
    __shared__ __attribute__((aligned(16))) float outp[256 * COUT];       // results, [pixel of the tile][cout]
    const int tid = threadIdx.x;
    for (int i = tid; i < K * COUT; i += 256) {
        const int k = i / COUT, o = i - k * COUT;
        const int tap = k / CIN, c = k - tap * CIN;
        wl[i] = a.weight[((long long)o * 9 + tap) * a.cin_pad + c];
    }
    const int tiles_x = (a.W + 15) / 16, tiles_y = (a.H + 15) / 16;
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int img = b / tiles_y;
    const int ox0 = tx * 16, oy0 = ty * 16;
    for (int i = tid; i < 18 * 18; i += 256) {
        const int hy = i / 18, hx = i - hy * 18;
        const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
            const float* p = a.in + (((long long)img * a.H + iy) * a.W + ix) * a.in_ld;
            if (CIN == 4 || (a.in_ld & 3) == 0) {       // 16-byte pixel records (NHWC4): one load
                v = *reinterpret_cast<const f32x4*>(p);
            } else {
                v.x = p[0];
                if (CIN > 1) v.y = p[1];
                if (CIN > 2) v.z = p[2];
            }
        }
        *reinterpret_cast<f32x4*>(tile + 4 * i) = v;
    }
    __syncthreads();
    const int px = tid & 15, py = tid >> 4;
    float x[K];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(tile + 4 * ((py + t / 3) * 18 + px + t % 3));
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int c = 0; c < CIN; ++c) x[t * CIN + c] = e[c];
    }
#pragma unroll
    for (int o4 = 0; o4 < CO4; ++o4) {
        f32x4 acc = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + 4 * o4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x2 a01 = acc.xy, a23 = acc.zw;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(wl + k * COUT + 4 * o4);     // same address in every lane: broadcast
            // explicit packed fused multiply-adds (v_pk_fma_f32: half the instructions of v_pk_mul + v_pk_add, which in turn beat four
            // scalar fmaf, 0.24 against 0.36 ms; always fused, so the result does not depend on how the loop is scheduled)
            const f32x2 xx = {x[k], x[k]};
            a01 = __builtin_elementwise_fma(xx, w.xy, a01);
            a23 = __builtin_elementwise_fma(xx, w.zw, a23);
        }
        acc = (f32x4){a01.x, a01.y, a23.x, a23.y};
        if (a.prelu) {
            const f32x4 p = *reinterpret_cast<const f32x4*>(a.prelu + 4 * o4);
            acc.x = acc.x > 0.f ? acc.x : p.x * acc.x;
            acc.y = acc.y > 0.f ? acc.y : p.y * acc.y;
            acc.z = acc.z > 0.f ? acc.z : p.z * acc.z;
            acc.w = acc.w > 0.f ? acc.w : p.w * acc.w;
        }
        *reinterpret_cast<f32x4*>(outp + tid * COUT + 4 * o4) = acc;
    }
    // Store through LDS in memory order: a thread's COUT floats are COUT*4 bytes apart from its neighbour's, so direct 16-byte
    // stores touch every line COUT/4 times with partial sectors (PMC: 2.2x the output bytes written).  Each wave owns 4 tile rows
    // of 16 pixels; piece q = 16 bytes of the wave's [4][16 pixels][COUT] image, written out 1 KiB per instruction.
    const int lane = tid & 63, wave = tid >> 6;
    if (a.out_ld == COUT) {
#pragma unroll
        for (int k = 0; k < CO4; ++k) {
            const int q = k * 64 + lane;                        // 0 .. 4*16*CO4
            const int row = q / (16 * CO4);
            const int off = q - row * (16 * CO4);               // 16-byte piece inside the row: pixel off / CO4, channels 4*(off % CO4)
            const int oy = oy0 + 4 * wave + row, ox = ox0 + off / CO4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(outp + (wave * 64 + row * 16) * COUT + 4 * off);
            if (oy < a.H && ox < a.W)
                *reinterpret_cast<f32x4*>(a.out + (((long long)img * a.H + oy) * a.W + ox0) * COUT + 4 * off) = v;
        }
    } else {                                                   // channel-slice output view: pixel records are not contiguous
        const int oy = oy0 + py, ox = ox0 + px;
        if (oy < a.H && ox < a.W) {
            float* orow = a.out + (((long long)img * a.H + oy) * a.W + ox) * a.out_ld;
#pragma unroll
            for (int o4 = 0; o4 < CO4; ++o4) *reinterpret_cast<f32x4*>(orow + 4 * o4) = *reinterpret_cast<const f32x4*>(outp + tid * COUT + 4 * o4);
        }
    }
}

template <int CIN, int CO4>
int launch_small(const GemmDev& d, hipStream_t s) {
    const long long blocks = (long long)(d.M / ((long long)d.H * d.W)) * ((d.W + 15) / 16) * ((d.H + 15) / 16);
    ATMVFI_REQUIRE(blocks < (1ll << 31), ATMVFI_EINVAL, "conv3x3_small: grid too large");
    hipLaunchKernelGGL((conv3x3_small_kernel<CIN, CO4>), dim3((unsigned)blocks), dim3(256), 0, s, d);
    return atmvfi::check_launch("conv3x3_small");
}

}  // namespace

// true + launched when the layer qualifies (3x3 s1 p1 d1, Cin <= 4, Cout in {16, 24, 32}, plain epilogue); false otherwise
bool atmvfi::try_launch_conv3x3_small(const GemmDev& d, int kh, int* rc, hipStream_t s) {
    if (d.mode != ATMVFI_GEMM_CONV || kh != 3 || d.kw != 3 || d.stride != 1 || d.pad != 1 || d.dil != 1 || d.in_prelu || d.residual ||
        d.out_row_map || d.Cin > 4 || d.Cin < 1 || d.Ho != d.H || d.Wo != d.W)
        return false;
    if (d.Cin != 3) return false;                       // the only shape the network has; others stay on the MFMA engine
    switch (d.Cout) {
        case 16: *rc = launch_small<3, 4>(d, s); return true;
        case 24: *rc = launch_small<3, 6>(d, s); return true;
        case 32: *rc = launch_small<3, 8>(d, s); return true;
        default: return false;
    }
}

// HBM-bound kernels of the ATM-VFI hot path for gfx950: LayerNorm (+window gather), depth-wise
// 3x3 + GELU, backward bilinear warps, fused warp/blend synthesis, align_corners resize,
// frame packing, final residual.  All are one-pass, 16-byte-per-lane where the layout allows.
#include "common.h"

#include <math.h>
#include <stdlib.h>

namespace {

// ---------------------------------------------------------------------------------------
// LayerNorm: one wavefront per token row, row kept in L1 between the three sweeps.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ in, int in_ld, long long in_gstride,
                                                        int in_rpg, const int* __restrict__ src_map,
                                                        const RowSink out,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        long long rows, int C) {
    fp16_saturate_on();
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int c4n = C >> 2;
    long long src = src_map ? src_map[row] : row;
    if (src < 0) {   // zero-padded token: LayerNorm(0) == beta exactly
        for (int i = lane; i < c4n; i += 64)
            sink_store4(out, row, 4 * i, *reinterpret_cast<const f32x4*>(beta + 4 * i));
        return;
    }
    const float* irow = in + ((in_rpg > 0) ? (src / in_rpg) * in_gstride + (src % in_rpg) * (long long)in_ld
                                           : src * (long long)in_ld);
    if (c4n <= 128) {
        // rows of up to 512 channels (every LayerNorm of both variants): the lane's one or two vectors stay in registers over the
        // three sweeps (same arithmetic in the same order as the loops below)
        const bool has1 = lane + 64 < c4n, has0 = lane < c4n;
        const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
        const f32x4 v0 = has0 ? *reinterpret_cast<const f32x4*>(irow + 4 * lane) : z;
        const f32x4 v1 = has1 ? *reinterpret_cast<const f32x4*>(irow + 4 * (lane + 64)) : z;
        float sum = 0.f;
        if (has0) sum += (v0.x + v0.y) + (v0.z + v0.w);
        if (has1) sum += (v1.x + v1.y) + (v1.z + v1.w);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float mean = sum / (float)C;
        float sq = 0.f;
        if (has0) {
            const float a = v0.x - mean, b = v0.y - mean, c = v0.z - mean, d = v0.w - mean;
            sq += (a * a + b * b) + (c * c + d * d);
        }
        if (has1) {
            const float a = v1.x - mean, b = v1.y - mean, c = v1.z - mean, d = v1.w - mean;
            sq += (a * a + b * b) + (c * c + d * d);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
        const float rstd = 1.0f / sqrtf(sq / (float)C + 1e-5f);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = lane + 64 * k;
            if (k == 0 ? has0 : has1) {
                const f32x4 v = k == 0 ? v0 : v1;
                const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + 4 * i);
                const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + 4 * i);
                f32x4 y;
                y.x = (v.x - mean) * rstd * gm.x + bt.x;
                y.y = (v.y - mean) * rstd * gm.y + bt.y;
                y.z = (v.z - mean) * rstd * gm.z + bt.z;
                y.w = (v.w - mean) * rstd * gm.w + bt.w;
                sink_store4(out, row, 4 * i, y);
            }
        }
        return;
    }
    float sum = 0.f;
    for (int i = lane; i < c4n; i += 64) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(irow + 4 * i);
        sum += (v.x + v.y) + (v.z + v.w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)C;
    float sq = 0.f;
    for (int i = lane; i < c4n; i += 64) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(irow + 4 * i);
        const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
        sq += (a * a + b * b) + (c * c + d * d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = 1.0f / sqrtf(sq / (float)C + 1e-5f);
    for (int i = lane; i < c4n; i += 64) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(irow + 4 * i);
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + 4 * i);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + 4 * i);
        f32x4 y;
        y.x = (v.x - mean) * rstd * gm.x + bt.x;
        y.y = (v.y - mean) * rstd * gm.y + bt.y;
        y.z = (v.z - mean) * rstd * gm.z + bt.z;
        y.w = (v.w - mean) * rstd * gm.w + bt.w;
        sink_store4(out, row, 4 * i, y);
    }
}

// ---------------------------------------------------------------------------------------
// depth-wise 3x3 (pad 1) + bias + exact GELU, NHWC, 4 channels per lane
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dwconv_gelu_kernel(const float* __restrict__ in, int in_ld,
                                                          const RowSink out,
                                                          const float* __restrict__ w9, const float* __restrict__ bias,
                                                          int N, int H, int W, int C) {
    fp16_saturate_on();
    const int c4n = C >> 2;
    const long long total = (long long)N * H * W * c4n;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c4n) << 2;
        const long long pix = idx / c4n;
        const int x = (int)(pix % W);
        const int y = (int)((pix / W) % H);
        f32x4 acc = *reinterpret_cast<const f32x4*>(bias + c);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = y + ky - 1;
            if ((unsigned)yy >= (unsigned)H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = x + kx - 1;
                if ((unsigned)xx >= (unsigned)W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(in + (pix + (long long)(ky - 1) * W + (kx - 1)) * in_ld + c);
                const f32x4 wv = *reinterpret_cast<const f32x4*>(w9 + (ky * 3 + kx) * C + c);
                acc.x += v.x * wv.x;
                acc.y += v.y * wv.y;
                acc.z += v.z * wv.z;
                acc.w += v.w * wv.w;
            }
        }
        f32x4 o;
        o.x = gelu_erf2(acc.x);
        o.y = gelu_erf2(acc.y);
        o.z = gelu_erf2(acc.z);
        o.w = gelu_erf2(acc.w);
        sink_store4(out, pix, c, o);
    }
}

// Sliding-window variant: a 256-thread block owns 16 x-positions x 64 channels and walks RS rows down
// the image keeping the 3x3 window of float4 in registers: 3 loads per output instead of 9, the x
// neighbours are shared inside the block (L1), and every input row is fetched from HBM once per
// RS-row strip ((RS+2)/RS over-read) instead of once per tap from three rows 1.5 MB apart
// (measured on the per-pixel kernel: 7x the algorithmic read traffic, profiles/r01_pmc_hbm_traffic.json).
template <int RS>
__global__ __launch_bounds__(256) void dwconv_gelu_rows_kernel(const float* __restrict__ in, int in_ld,
                                                               const RowSink out,
                                                               const float* __restrict__ w9, const float* __restrict__ bias,
                                                               int N, int H, int W, int C, int xblocks, int cblocks, int strips) {
    fp16_saturate_on();
    int bid = blockIdx.x;
    const int cb = bid % cblocks; bid /= cblocks;
    const int xb = bid % xblocks; bid /= xblocks;
    const int sb = bid % strips;
    const int n = bid / strips;
    const int c = (cb * 16 + (threadIdx.x & 15)) << 2;
    const int x = xb * 16 + (threadIdx.x >> 4);
    if (x >= W) return;
    const int y0 = sb * RS;
    f32x4 wv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const f32x4*>(w9 + t * C + c);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + c);
    const bool xm_ok = x > 0, xp_ok = x + 1 < W;
    const float* base = in + ((long long)n * H * W + x) * in_ld + c;
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
    // Borders: a column outside the image gets ZERO WEIGHTS (its loads are clamped to a valid address, and finite x * 0 adds exactly
    // nothing; the blocks on the left / right image edge -- a block-uniform case -- also select a zero VALUE at the use, so that an
    // Inf / NaN centre pixel in column 0 or W - 1 adds nothing either, as zero padding does: ADVICE round 4), a row outside the image is a block-uniform case and becomes zeros without a load -- so a loaded value is first touched by
    // the FMAs, and row y + 2 can be requested one iteration ahead: the wait for it then sits behind a whole row of arithmetic and behind
    // this row's stores in program order (vmcnt retires in order: waiting for a load issued AFTER the previous row's stores also waited
    // for those stores to be acknowledged -- once per row, with the load latency on top).
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        if (!xm_ok) wv[ky * 3 + 0] = zero;
        if (!xp_ok) wv[ky * 3 + 2] = zero;
    }
    // A strip whose RS + 2 input rows all lie inside the image (all but the first and the last strip of a map) runs without any row test:
    // straight-line code, so that hipcc's waits are counted ones (at a control-flow join it waits for vmcnt(0)).
    auto run = [&](auto interior_tag, auto xedge_tag) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;
        constexpr bool XEDGE = decltype(xedge_tag)::value;
        f32x4 win[4][3];        // [row slot][x-1, x, x+1]
        const long long xm = xm_ok ? in_ld : 0, xp = xp_ok ? in_ld : 0;
        auto load_row = [&](int y, f32x4 (&dst)[3]) {
            if (INTERIOR || (unsigned)y < (unsigned)H) {                        // (uniform over the block)
                const float* p = base + (long long)y * W * in_ld;
                dst[0] = *reinterpret_cast<const f32x4*>(p - xm);
                dst[1] = *reinterpret_cast<const f32x4*>(p);
                dst[2] = *reinterpret_cast<const f32x4*>(p + xp);
            } else {
                dst[0] = zero; dst[1] = zero; dst[2] = zero;
            }
        };
        load_row(y0 - 1, win[0]);
        load_row(y0, win[1]);
        load_row(y0 + 1, win[2]);
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            const int y = y0 + r;
            if (r + 1 < RS) load_row(y + 2, win[(r + 3) % 4]);
            if (INTERIOR || y < H) {
                // explicit packed fused multiply-adds (v_pk_fma_f32: 18 instructions per 4-channel output instead of 36 mul + 36 add
                // under -ffp-contract=off; always fused, so the result does not depend on how the loop is peeled)
                f32x2 a01 = bv.xy, a23 = bv.zw;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const f32x4* row = win[(r + ky) % 4];
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        f32x4 v = row[kx];
                        const f32x4 wt = wv[ky * 3 + kx];
                        if (XEDGE && kx != 1) v = (kx == 0 ? xm_ok : xp_ok) ? v : zero;
                        a01 = __builtin_elementwise_fma(v.xy, wt.xy, a01);
                        a23 = __builtin_elementwise_fma(v.zw, wt.zw, a23);
                    }
                }
                const f32x4 acc = {a01.x, a01.y, a23.x, a23.y};
                f32x4 o;
                o.x = gelu_erf2(acc.x);
                o.y = gelu_erf2(acc.y);
                o.z = gelu_erf2(acc.z);
                o.w = gelu_erf2(acc.w);
                sink_store4(out, ((long long)n * H + y) * W + x, c, o);
            }
        }
    };
    const bool xedge = xb == 0 || xb == xblocks - 1;
    if (y0 >= 1 && y0 + RS + 1 <= H) {
        if (xedge) run(std::true_type{}, std::true_type{});
        else run(std::true_type{}, std::false_type{});
    } else {
        if (xedge) run(std::false_type{}, std::true_type{});
        else run(std::false_type{}, std::false_type{});
    }
}

// LDS-DMA variant for the plane sink (the Mlp's dw-conv always feeds fc2's split planes).  The sliding-window kernel above keeps ONE
// row of prefetch in registers and runs 3 waves per SIMD (158 registers): ~12 KB of unique input in flight per CU, and it reaches
// 4.0 TB/s of algorithmic traffic where a plain fp32 -> planes copy (split_planes) reaches 5.6.  Here every WAVE owns 8 x-positions x
// 32 channels and a private ring of DW_RING input rows in LDS (10 columns x 128 B: its 8 columns + the two neighbours), filled by
// global_load_lds_dwordx4 DW_RING rows ahead: no register holds a value in flight, the fetch depth is the ring's, nothing is shared
// between waves (no barrier; a wave's own counted vmcnt orders its reads behind its DMA).  Per input row: two DMA instructions (1 KiB +
// 256 B on lanes 0-15), three ds_read_b128 per lane (the new row of the 3x3 window; the other two rows stay in registers), and per
// output row the two 8-byte plane stores -- the vmcnt immediates count all of them (vmcnt retires in issue order on gfx9).
// Every strip has exactly RS output rows (the launcher picks RS | H, or lets the last strip overlap the one before: same values
// written twice), so the instruction stream -- and the counts -- are the same for every wave.  Arithmetic and its order are those of
// the sliding-window kernel: bit-identical results.
constexpr int DW_RING = 6;          // (5, 6 and 7 rows measure the same: tools/dw_ab.sh, profiles/r05_dwconv_dma_ab.txt)
constexpr int DW_SLOT = 10 * 128;
template <int OFF>
__device__ __forceinline__ void lds_read4f(f32x4& d, unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// VM operations a wave has issued behind the DMA of input row j when it comes to wait for that row at the top of step j.  Program order:
// prologue DMA(0) .. DMA(DW_RING - 1); step i: [DMA(i + DW_RING) if that row exists], [the two stores of output row i - 2 if i >= 2].
template <int RS>
constexpr int dw_behind(int j) {
    const int rows = RS + 2;
    int n = 0;
    if (j < DW_RING) {
        n += 2 * ((DW_RING < rows ? DW_RING : rows) - 1 - j);
        for (int i = 0; i < j; ++i) n += (i + DW_RING < rows ? 2 : 0) + (i >= 2 ? 2 : 0);
    } else {
        const int i0 = j - DW_RING;          // the step that issued DMA(j)
        n += i0 >= 2 ? 2 : 0;
        for (int i = i0 + 1; i < j; ++i) n += (i + DW_RING < rows ? 2 : 0) + (i >= 2 ? 2 : 0);
    }
    return n;
}
template <int RS>
__global__ __launch_bounds__(256) void dwconv_gelu_dma_kernel(const float* __restrict__ in, int in_ld, _Float16* __restrict__ out_hi,
                                                              _Float16* __restrict__ out_lo, long long plane_rows,
                                                              const float* __restrict__ w9, const float* __restrict__ bias, int N, int H,
                                                              int W, int C, int xgroups, int cblocks, int strips) {
    static_assert(RS + 2 >= DW_RING, "the prologue fills the whole ring");
    fp16_saturate_on();
    extern __shared__ __attribute__((aligned(16))) unsigned char dw_smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // unit = one wave's strip: x-group fastest (the four waves of a block are neighbours in x and share their halo columns in L1),
    // then the 32-channel block (adjacent 128-byte lines), the strip, the image
    unsigned u = blockIdx.x * 4u + (unsigned)wave;          // (the launcher keeps the unit count below 2^31)
    const int xg = (int)(u % (unsigned)xgroups); u /= (unsigned)xgroups;
    const int cbk = (int)(u % (unsigned)cblocks); u /= (unsigned)cblocks;
    const int sb = (int)(u % (unsigned)strips);
    const int n = (int)(u / (unsigned)strips);
    if (n >= N) return;
    int y0 = sb * RS;
    if (y0 + RS > H) y0 = H - RS;          // (the last strip of a map whose height RS does not divide overlaps the previous one)
    const int c = cbk * 32 + ((lane & 7) << 2);
    const int x = xg * 8 + (lane >> 3);
    unsigned char* ring = dw_smem + wave * (DW_RING * DW_SLOT);
    const unsigned rd = lds_offset(ring) + (unsigned)(lane * 16);      // column (lane >> 3) of a slot = x - 1; + 128 k: x - 1 + k

    f32x4 wv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const f32x4*>(w9 + t * C + c);
    f32x4 bv = *reinterpret_cast<const f32x4*>(bias + c);
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool xm_ok = x > 0, xp_ok = x + 1 < W;
    // borders as in the sliding-window kernel: a column outside the image has zero weights and a clamped (valid) load address; the waves
    // on the image's left / right edge also select a zero VALUE; a row outside the image (row -1 of the first strip, row H of the last)
    // is loaded from a clamped address and replaced by zeros
    // DMA sources: piece A = columns 0..7 of the slot (x-group's x - 1 .. + 6), piece B = columns 8, 9 on lanes 0..15
    auto clampx = [&](int xx) { return xx < 0 ? 0 : (xx >= W ? W - 1 : xx); };
    const int xa = clampx(xg * 8 - 1 + (lane >> 3));
    const int xb = clampx(xg * 8 + 7 + ((lane >> 3) & 1));
    const long long img = (long long)n * H * W;
    const float* src_a = in + (img + xa) * in_ld + c;
    const float* src_b = in + (img + xb) * in_ld + c;
    const long long row_stride = (long long)W * in_ld;
    auto issue_row = [&](int j) {          // input row j of the strip = image row y0 - 1 + j -> slot j % DW_RING
        int y = y0 - 1 + j;
        y = y < 0 ? 0 : (y >= H ? H - 1 : y);
        unsigned char* dst = ring + (j % DW_RING) * DW_SLOT;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_a + y * row_stride),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        // piece B on lanes 0..15 only, as ONE instruction whatever the compiler makes of the code around it (an `if (lane < 16)` around the
        // builtin was tail-merged with the next row's piece A in the 8-row instance: four DMA instructions where the counts assume
        // three).  EXEC is all ones here (wave-uniform control flow, full waves); M0 is written in the statement that uses it.
        unsigned keep_m0;
        unsigned long long keep_exec;
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b64 %1, exec\n\t"
                     "s_mov_b32 m0, %3\n\t"
                     "s_mov_b64 exec, 0xffff\n\t"
                     "global_load_lds_dwordx4 %2, off\n\t"
                     "s_mov_b64 exec, %1\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep_m0), "=&s"(keep_exec)
                     : "v"(src_b + y * row_stride), "s"(lds_offset(dst + 1024))
                     : "memory");
    };
    const bool top = y0 == 0, bottom = y0 + RS == H;
    const long long orow0 = img + (long long)y0 * W + x;
    const long long obase = ((long long)(c >> 5) * plane_rows) * 32 + (c & 31);
    // the whole ring goes in flight behind the weight loads and before their first use (hipcc counts the DMAs in its wait for the weights)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < DW_RING; ++j) issue_row(j);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        if (!xm_ok) wv[ky * 3 + 0] = zero;
        if (!xp_ok) wv[ky * 3 + 2] = zero;
    }

    auto run = [&](auto xedge_tag) {
        constexpr bool XEDGE = decltype(xedge_tag)::value;
        f32x4 win[3][3];
        static_for<0, RS + 2>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int slot = (j % DW_RING) * DW_SLOT;
            wait_vmcnt<dw_behind<RS>(j)>();
            lds_read4f<slot>(win[j % 3][0], rd);
            lds_read4f<slot + 128>(win[j % 3][1], rd);
            lds_read4f<slot + 256>(win[j % 3][2], rd);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(win[j % 3][0]), "+v"(win[j % 3][1]), "+v"(win[j % 3][2])::"memory");
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (j + DW_RING < RS + 2) issue_row(j + DW_RING);
            __builtin_amdgcn_sched_barrier(0);       // (the counts above assume this order: DMA, then the step's stores)
            if constexpr (j == 0) {
                if (top) { win[0][0] = zero; win[0][1] = zero; win[0][2] = zero; }
            }
            if constexpr (j == RS + 1) {
                if (bottom) { win[j % 3][0] = zero; win[j % 3][1] = zero; win[j % 3][2] = zero; }
            }
            if constexpr (j >= 2) {
                constexpr int r = j - 2;
                f32x2 a01 = bv.xy, a23 = bv.zw;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const f32x4* row = win[(r + ky) % 3];
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        f32x4 v = row[kx];
                        const f32x4 wt = wv[ky * 3 + kx];
                        if (XEDGE && kx != 1) v = (kx == 0 ? xm_ok : xp_ok) ? v : zero;
                        a01 = __builtin_elementwise_fma(v.xy, wt.xy, a01);
                        a23 = __builtin_elementwise_fma(v.zw, wt.zw, a23);
                    }
                }
                f32x4 o;
                o.x = gelu_erf2(a01.x);
                o.y = gelu_erf2(a01.y);
                o.z = gelu_erf2(a23.x);
                o.w = gelu_erf2(a23.y);
                f16x2 h0, l0, h1, l1;
                split_pair((f32x2){o.x, o.y}, h0, l0);
                split_pair((f32x2){o.z, o.w}, h1, l1);
                const f16x4 h = {h0.x, h0.y, h1.x, h1.y}, l = {l0.x, l0.y, l1.x, l1.y};
                const long long off = obase + (orow0 + (long long)r * W) * 32;
                // (x >= W: lanes of the last x-group past the image; the store instructions are still issued -- some lane of the group is inside)
                if (x < W) {
                    *reinterpret_cast<f16x4*>(out_hi + off) = h;
                    *reinterpret_cast<f16x4*>(out_lo + off) = l;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };
    if (xg == 0 || xg == xgroups - 1) run(std::true_type{});
    else run(std::false_type{});
}

__global__ void pack_dw_kernel(const float* __restrict__ src, float* __restrict__ dst, int C) {
    fp16_saturate_on();
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < 9 * C) {
        const int tap = idx / C, c = idx - tap * C;
        dst[idx] = src[c * 9 + tap];
    }
}

// ---------------------------------------------------------------------------------------
// Sampling coordinates exactly as the reference computes them (flow_warp.py:35-36 normalise,
// then grid_sample(align_corners=True) un-normalises): the round trip is kept so that the
// tap positions are bit-identical to the reference's.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float ref_coord(float p, int size) {
    const float g = 2.0f * p / (float)(size - 1) - 1.0f;
    return ((g + 1.0f) / 2.0f) * (float)(size - 1);
}

struct Taps {
    int x0, y0;
    float w00, w01, w10, w11;   // (y0,x0) (y0,x1) (y1,x0) (y1,x1); zero when the tap is outside
    bool in00, in01, in10, in11;
};

__device__ __forceinline__ Taps make_taps(float px, float py, int W, int H) {
    Taps t;
    const float ix = ref_coord(px, W), iy = ref_coord(py, H);
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    // clamp before the int conversion so wild flows cannot overflow
    const float cx = fminf(fmaxf(fx0, -2.0f), (float)W + 1.0f), cy = fminf(fmaxf(fy0, -2.0f), (float)H + 1.0f);
    t.x0 = (int)cx;
    t.y0 = (int)cy;
    const float ax = ix - fx0, ay = iy - fy0;   // weight of the +1 neighbour
    const bool far = (cx != fx0) || (cy != fy0) || !(ix == ix) || !(iy == iy);
    t.in00 = !far && t.x0 >= 0 && t.x0 < W && t.y0 >= 0 && t.y0 < H;
    t.in01 = !far && t.x0 + 1 >= 0 && t.x0 + 1 < W && t.y0 >= 0 && t.y0 < H;
    t.in10 = !far && t.x0 >= 0 && t.x0 < W && t.y0 + 1 >= 0 && t.y0 + 1 < H;
    t.in11 = !far && t.x0 + 1 >= 0 && t.x0 + 1 < W && t.y0 + 1 >= 0 && t.y0 + 1 < H;
    t.w00 = (1.0f - ax) * (1.0f - ay);
    t.w01 = ax * (1.0f - ay);
    t.w10 = (1.0f - ax) * ay;
    t.w11 = ax * ay;
    return t;
}

__device__ __forceinline__ float sample_plane(const float* __restrict__ p, const Taps& t, int W) {
    float v = 0.f;
    const float* b = p + (long long)t.y0 * W + t.x0;
    if (t.in00) v += b[0] * t.w00;
    if (t.in01) v += b[1] * t.w01;
    if (t.in10) v += b[W] * t.w10;
    if (t.in11) v += b[W + 1] * t.w11;
    return v;
}

__global__ __launch_bounds__(256) void flow_warp_planar_kernel(const float* __restrict__ src, const float* __restrict__ flow,
                                                               long long flow_bstride, int flow_pstride, int flow_cstride,
                                                               float* __restrict__ dst, int B, int C, int H, int W) {
    fp16_saturate_on();
    const long long hw = (long long)H * W;
    const long long total = (long long)B * hw;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(idx / hw);
        const long long pix = idx - (long long)b * hw;
        const int y = (int)(pix / W), x = (int)(pix - (long long)y * W);
        const float* fp = flow + b * flow_bstride + pix * flow_pstride;
        const Taps t = make_taps((float)x + fp[0], (float)y + fp[flow_cstride], W, H);
        for (int c = 0; c < C; ++c)
            dst[((long long)b * C + c) * hw + pix] = sample_plane(src + ((long long)b * C + c) * hw, t, W);
    }
}

// The non-default forms of the reference's flow_warp (flow_warp.py:50-60 -> bilinear_sample :26-47): padding_mode 'border' /
// 'reflection' (grid_sample's coordinate maps for align_corners=True: clip to [0, size-1]; reflect about 0 and size-1, then clip)
// and return_mask (the normalised coordinate inside [-1, 1] on both axes, evaluated on the reference's own fp32 expression
// 2 p / (size - 1) - 1, so that a coordinate a rounding below zero is "inside" exactly when the reference says so).  Off the hot path:
// one lane per pixel, direct gathers.  mode: 0 zeros, 1 border, 2 reflection.
__device__ __forceinline__ float pad_coord(float c, int size, int mode) {
    if (mode == 0) return c;
    const float hi = (float)(size - 1);
    if (mode == 2) {                                         // reflect_coordinates(c, 0, 2 (size - 1)) of ATen's GridSampler.h
        if (size == 1) c = 0.f;
        else {
            const float a = fabsf(c), extra = fmodf(a, hi);
            const int flips = (int)floorf(a / hi);
            c = (flips & 1) ? hi - extra : extra;
        }
    }
    return fminf(hi, fmaxf(c, 0.f));                         // clip_coordinates
}
__global__ __launch_bounds__(256) void flow_warp_ex_kernel(const float* __restrict__ src, const float* __restrict__ flow, float* __restrict__ dst,
                                                           unsigned char* __restrict__ mask, int B, int C, int H, int W, int mode) {
    fp16_saturate_on();
    const long long hw = (long long)H * W;
    const long long total = (long long)B * hw;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(idx / hw);
        const long long pix = idx - (long long)b * hw;
        const int y = (int)(pix / W), x = (int)(pix - (long long)y * W);
        const float* fp = flow + (long long)b * 2 * hw + pix;
        const float px = (float)x + fp[0], py = (float)y + fp[hw];
        const float gx = 2.0f * px / (float)(W - 1) - 1.0f, gy = 2.0f * py / (float)(H - 1) - 1.0f;
        if (mask) mask[idx] = (gx >= -1.0f) && (gy >= -1.0f) && (gx <= 1.0f) && (gy <= 1.0f);
        const float ix = pad_coord(((gx + 1.0f) / 2.0f) * (float)(W - 1), W, mode), iy = pad_coord(((gy + 1.0f) / 2.0f) * (float)(H - 1), H, mode);
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        const float cx = fminf(fmaxf(fx0, -2.0f), (float)W + 1.0f), cy = fminf(fmaxf(fy0, -2.0f), (float)H + 1.0f);
        Taps t;
        t.x0 = (int)cx;
        t.y0 = (int)cy;
        const float ax = ix - fx0, ay = iy - fy0;
        const bool far = (cx != fx0) || (cy != fy0) || !(ix == ix) || !(iy == iy);
        t.in00 = !far && t.x0 >= 0 && t.x0 < W && t.y0 >= 0 && t.y0 < H;
        t.in01 = !far && t.x0 + 1 >= 0 && t.x0 + 1 < W && t.y0 >= 0 && t.y0 < H;
        t.in10 = !far && t.x0 >= 0 && t.x0 < W && t.y0 + 1 >= 0 && t.y0 + 1 < H;
        t.in11 = !far && t.x0 + 1 >= 0 && t.x0 + 1 < W && t.y0 + 1 >= 0 && t.y0 + 1 < H;
        t.w00 = (1.0f - ax) * (1.0f - ay);
        t.w01 = ax * (1.0f - ay);
        t.w10 = (1.0f - ax) * ay;
        t.w11 = ax * ay;
        for (int c = 0; c < C; ++c)
            dst[((long long)b * C + c) * hw + pix] = sample_plane(src + ((long long)b * C + c) * hw, t, W);
    }
}

// flow_warp of a planar image by a planar flow AND the x2 up-sampling of that flow to the next finer level (upsample_flow,
// network_base.py:11-18: bilinear, align_corners=True, values x 2) in one launch: the global flow's walk down the image pyramid
// (network_base.py:468-485) is four warps and three up-samplings in a chain.  The two halves of the index space are independent
// and use the arithmetic of flow_warp_planar_kernel / resize_ac_kernel unchanged: bit-identical to the two launches.
__global__ __launch_bounds__(256) void flow_warp_up2_kernel(const float* __restrict__ src, const float* __restrict__ flow, float* __restrict__ dst,
                                                            float* __restrict__ flow_up, int B, int C, int H, int W) {
    fp16_saturate_on();
    const long long hw = (long long)H * W;
    const long long nwarp = (long long)B * hw;
    const int Ho = 2 * H, Wo = 2 * W;
    const long long ohw = (long long)Ho * Wo;
    const long long nup = (long long)B * 2 * ohw;
    const float sh = (float)(H - 1) / (float)(Ho - 1), sw = (float)(W - 1) / (float)(Wo - 1);
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < nwarp + nup; idx += (long long)gridDim.x * blockDim.x) {
        if (idx < nwarp) {
            const int b = (int)(idx / hw);
            const long long pix = idx - (long long)b * hw;
            const int y = (int)(pix / W), x = (int)(pix - (long long)y * W);
            const float* fp = flow + (long long)b * 2 * hw + pix;
            const Taps t = make_taps((float)x + fp[0], (float)y + fp[hw], W, H);
            for (int c = 0; c < C; ++c)
                dst[((long long)b * C + c) * hw + pix] = sample_plane(src + ((long long)b * C + c) * hw, t, W);
        } else {
            const long long e = idx - nwarp;
            const int p = (int)(e / ohw);                    // plane b * 2 + c
            const long long pix = e - (long long)p * ohw;
            const int oy = (int)(pix / Wo), ox = (int)(pix - (long long)oy * Wo);
            const float ry = sh * (float)oy, rx = sw * (float)ox;
            const int y0 = (int)ry, x0 = (int)rx;
            const int yp = (y0 < H - 1) ? 1 : 0, xp = (x0 < W - 1) ? 1 : 0;
            const float ly = ry - (float)y0, lx = rx - (float)x0;
            const float hy = 1.0f - ly, hx = 1.0f - lx;
            const float* s = flow + (long long)p * hw + (long long)y0 * W + x0;
            const float v = hy * (hx * s[0] + lx * s[xp]) + ly * (hx * s[yp * W] + lx * s[yp * W + xp]);
            flow_up[e] = v * 2.0f;
        }
    }
}

__global__ __launch_bounds__(256) void flow_warp_nhwc_kernel(const float* __restrict__ src, int src_ld, long long src_bstride,
                                                             const float* __restrict__ flow, long long flow_bstride,
                                                             int flow_pstride, int flow_cstride, float* __restrict__ dst,
                                                             int dst_ld, long long dst_bstride, int B, int C, int H, int W) {
    fp16_saturate_on();
    const int c4n = C >> 2;
    const long long hw = (long long)H * W;
    const long long total = (long long)B * hw * c4n;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c4n) << 2;
        const long long bp = idx / c4n;
        const int b = (int)(bp / hw);
        const long long pix = bp - (long long)b * hw;
        const int y = (int)(pix / W), x = (int)(pix - (long long)y * W);
        const float* fp = flow + b * flow_bstride + pix * flow_pstride;
        const Taps t = make_taps((float)x + fp[0], (float)y + fp[flow_cstride], W, H);
        const float* sb = src + b * src_bstride + ((long long)t.y0 * W + t.x0) * src_ld + c;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (t.in00) { const f32x4 v = *reinterpret_cast<const f32x4*>(sb); acc += v * t.w00; }
        if (t.in01) { const f32x4 v = *reinterpret_cast<const f32x4*>(sb + src_ld); acc += v * t.w01; }
        if (t.in10) { const f32x4 v = *reinterpret_cast<const f32x4*>(sb + (long long)W * src_ld); acc += v * t.w10; }
        if (t.in11) { const f32x4 v = *reinterpret_cast<const f32x4*>(sb + (long long)(W + 1) * src_ld); acc += v * t.w11; }
        *reinterpret_cast<f32x4*>(dst + b * dst_bstride + pix * dst_ld + c) = acc;
    }
}

// ---------------------------------------------------------------------------------------
// fused synthesis at one pyramid level
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void warp_blend_kernel(const float* __restrict__ im0, const float* __restrict__ im1,
                                                         const float* __restrict__ motion, int motion_ld, long long motion_bstride,
                                                         float* __restrict__ i0w, float* __restrict__ i1w, float* __restrict__ it,
                                                         float* __restrict__ f0o, float* __restrict__ f1o,
                                                         float* __restrict__ m1o, float* __restrict__ m2o,
                                                         const float* __restrict__ orig0, const float* __restrict__ orig1,
                                                         float* __restrict__ pack15, int pack_ld, _Float16* __restrict__ pack_hi,
                                                         _Float16* __restrict__ pack_lo, long long pack_rows, int pack_c0, int B, int H,
                                                         int W) {
    fp16_saturate_on();
    const long long hw = (long long)H * W;
    const long long total = (long long)B * hw;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(idx / hw);
        const long long pix = idx - (long long)b * hw;
        const int y = (int)(pix / W), x = (int)(pix - (long long)y * W);
        const float* mp = motion + b * motion_bstride + pix * motion_ld;
        const float fx0 = mp[0], fy0 = mp[1], fx1 = mp[2], fy1 = mp[3];
        const float m1 = sigmoidf_(mp[4]);
        const float m2 = 1.0f - m1;
        const Taps t0 = make_taps((float)x + fx0, (float)y + fy0, W, H);
        const Taps t1 = make_taps((float)x + fx1, (float)y + fy1, W, H);
        float a[3], c[3], o[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const long long pl = ((long long)b * 3 + ch) * hw;
            a[ch] = sample_plane(im0 + pl, t0, W);
            c[ch] = sample_plane(im1 + pl, t1, W);
            o[ch] = m1 * a[ch] + m2 * c[ch];
            i0w[pl + pix] = a[ch];
            i1w[pl + pix] = c[ch];
            it[pl + pix] = o[ch];
        }
        if (f0o) {
            f0o[((long long)b * 2) * hw + pix] = fx0;
            f0o[((long long)b * 2 + 1) * hw + pix] = fy0;
            f1o[((long long)b * 2) * hw + pix] = fx1;
            f1o[((long long)b * 2 + 1) * hw + pix] = fy1;
        }
        if (m1o) {
            m1o[(long long)b * hw + pix] = m1;
            m2o[(long long)b * hw + pix] = m2;
        }
        if (pack15) {
            float* pp = pack15 + ((long long)b * hw + pix) * pack_ld;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const long long pl = ((long long)b * 3 + ch) * hw + pix;
                pp[ch] = orig0[pl];
                pp[3 + ch] = a[ch];
                pp[6 + ch] = orig1[pl];
                pp[9 + ch] = c[ch];
                pp[12 + ch] = o[ch];
            }
        }
        if (pack_hi) {
            // the same 15 values (+ one zero) as split planes at channels pack_c0 .. pack_c0 + 16: the refiner's first conv reads
            // its input as planes (atmvfi_conv3x3_planes)
            float v16[16];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const long long pl = ((long long)b * 3 + ch) * hw + pix;
                v16[ch] = orig0[pl];
                v16[3 + ch] = a[ch];
                v16[6 + ch] = orig1[pl];
                v16[9 + ch] = c[ch];
                v16[12 + ch] = o[ch];
            }
            v16[15] = 0.f;
            const RowSink sink{nullptr, 0, pack_hi, pack_lo, pack_rows};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                sink_store4(sink, (long long)b * hw + pix, pack_c0 + 4 * q, (f32x4){v16[4 * q], v16[4 * q + 1], v16[4 * q + 2], v16[4 * q + 3]});
        }
    }
}

// ---------------------------------------------------------------------------------------
// The same warps with LDS-STAGED SOURCE TILES (the north star's form of flow_warp.py:50-60): a workgroup owns a 32 x 8 tile of output
// pixels; it reduces the bounding box of its lanes' bilinear taps (cross-lane minima, one LDS atomic per wave), loads that box of every source plane with
// 16-byte row loads into LDS (up to 64 x 24 source pixels: flows of about +-15 px horizontally, +-8 vertically around the tile) and
// takes the four taps of every pixel from LDS -- ~5 coalesced dwordx4 loads per lane and source instead of 12 gathered dwords.  A tile
// whose flows reach further than the box gathers from global memory as the direct kernels do (a workgroup-uniform decision, per
// source).  Taps, weights and the order of the four multiply-adds are those of sample_plane: results are bit-identical to the direct
// kernels (tests/test_gpu_ops.py::test_tiled_warps_equal_direct_warps).  Needs W % 4 == 0 and 16-byte aligned planes.
// ---------------------------------------------------------------------------------------
constexpr int WT_W = 32, WT_H = 8;            // output tile of a workgroup (256 lanes)
constexpr int WB_W = 64, WB_H = 24;           // staged box per source plane, pixels
constexpr int WB_PLANE = WB_W * WB_H;

struct StagedBox {
    int ax0, y0, nv, h;      // columns ax0 .. ax0 + 4 nv - 1 (ax0 a multiple of 4), rows y0 .. y0 + h - 1
    bool ok;                 // it fits the LDS box (an empty box -- no live tap in the tile -- counts as staged)
};

__device__ __forceinline__ void box_reset(int* bx) {       // {x lo, x hi, y lo, y hi}
    bx[0] = 0x7fffffff; bx[1] = -0x7fffffff; bx[2] = 0x7fffffff; bx[3] = -0x7fffffff;
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) { const int u = __shfl_xor(v, o); v = u < v ? u : v; }
    return v;
}
__device__ __forceinline__ void box_add(int* bx, const Taps& t, bool live, int W, int H) {
    // the wave's extremes by cross-lane exchange, then one LDS atomic per wave and bound (64 lanes on one LDS word serialise)
    const bool any = live && (t.in00 || t.in01 || t.in10 || t.in11);      // some tap inside the image: x0 in [-1, W-1], y0 in [-1, H-1]
    const int xlo = any ? (t.x0 < 0 ? 0 : t.x0) : 0x7fffffff;
    const int xhi = any ? (t.x0 + 1 > W - 1 ? W - 1 : t.x0 + 1) : -0x7fffffff;
    const int ylo = any ? (t.y0 < 0 ? 0 : t.y0) : 0x7fffffff;
    const int yhi = any ? (t.y0 + 1 > H - 1 ? H - 1 : t.y0 + 1) : -0x7fffffff;
    const int a = wave_min_i32(xlo), b = -wave_min_i32(-xhi), c = wave_min_i32(ylo), d = -wave_min_i32(-yhi);
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&bx[0], a);
        atomicMax(&bx[1], b);
        atomicMin(&bx[2], c);
        atomicMax(&bx[3], d);
    }
}
__device__ __forceinline__ StagedBox box_get(const int* bx) {
    StagedBox b;
    if (bx[0] > bx[1]) { b.ax0 = 0; b.y0 = 0; b.nv = 0; b.h = 0; b.ok = true; return b; }
    b.ax0 = bx[0] & ~3;
    b.y0 = bx[2];
    b.nv = ((bx[1] - b.ax0) >> 2) + 1;
    b.h = bx[3] - bx[2] + 1;
    b.ok = b.nv <= WB_W / 4 && b.h <= WB_H;
    return b;
}
// nch planes (plane stride hw floats) of the box into tile[ch][row][WB_W]; all 256 lanes of the workgroup take part
__device__ __forceinline__ void box_stage(float* tile, const float* __restrict__ src, long long hw, int W, int nch, const StagedBox& b, int tid) {
    const int per = b.h * b.nv, total = nch * per;
    for (int i = tid; i < total; i += 256) {
        const int ch = i / per, rem = i - ch * per;
        const int r = rem / b.nv, v = rem - r * b.nv;
        const f32x4 q = *reinterpret_cast<const f32x4*>(src + ch * hw + (long long)(b.y0 + r) * W + b.ax0 + 4 * v);
        *reinterpret_cast<f32x4*>(tile + ch * WB_PLANE + r * WB_W + 4 * v) = q;
    }
}
__device__ __forceinline__ float sample_tile(const float* tile_plane, const Taps& t, const StagedBox& b) {
    float v = 0.f;
    const float* p = tile_plane + (t.y0 - b.y0) * WB_W + (t.x0 - b.ax0);
    if (t.in00) v += p[0] * t.w00;
    if (t.in01) v += p[1] * t.w01;
    if (t.in10) v += p[WB_W] * t.w10;
    if (t.in11) v += p[WB_W + 1] * t.w11;
    return v;
}
// tile origin of this workgroup: blockIdx.x = (b * tiles_y + ty) * tiles_x + tx
__device__ __forceinline__ void tile_pixel(int H, int W, int& b, int& x, int& y, bool& live) {
    const int tiles_x = (W + WT_W - 1) / WT_W, tiles_y = (H + WT_H - 1) / WT_H;
    const int t = blockIdx.x, tx = t % tiles_x, r = t / tiles_x;
    b = r / tiles_y;
    x = tx * WT_W + (threadIdx.x & (WT_W - 1));
    y = (r - b * tiles_y) * WT_H + (threadIdx.x >> 5);
    live = x < W && y < H;
}

__global__ __launch_bounds__(256) void flow_warp_tiled_kernel(const float* __restrict__ src, const float* __restrict__ flow,
                                                              long long flow_bstride, int flow_pstride, int flow_cstride,
                                                              float* __restrict__ dst, int B, int C, int H, int W) {
    fp16_saturate_on();
    __shared__ __attribute__((aligned(16))) float tile[3 * WB_PLANE];
    __shared__ int bx[4];
    const int tid = threadIdx.x;
    const long long hw = (long long)H * W;
    int b, x, y;
    bool live;
    tile_pixel(H, W, b, x, y, live);
    const long long pix = (long long)(live ? y : 0) * W + (live ? x : 0);
    if (tid == 0) box_reset(bx);
    const float* fp = flow + b * flow_bstride + pix * flow_pstride;
    const Taps t = make_taps((float)x + fp[0], (float)y + fp[flow_cstride], W, H);
    __syncthreads();
    box_add(bx, t, live, W, H);
    __syncthreads();
    const StagedBox sb = box_get(bx);
    for (int c0 = 0; c0 < C; c0 += 3) {
        const int nch = C - c0 < 3 ? C - c0 : 3;
        const float* sp = src + ((long long)b * C + c0) * hw;
        if (sb.ok) {
            if (c0) __syncthreads();                         // the previous group's taps have been read
            box_stage(tile, sp, hw, W, nch, sb, tid);
            __syncthreads();
        }
        if (live)
            for (int c = 0; c < nch; ++c)
                dst[((long long)b * C + c0 + c) * hw + pix] = sb.ok ? sample_tile(tile + c * WB_PLANE, t, sb) : sample_plane(sp + c * hw, t, W);
    }
}

// flow_warp_up2_kernel with the warp half on LDS-staged tiles: workgroups 0 .. ntiles-1 warp one 32 x 8 tile each (as
// flow_warp_tiled_kernel), the others up-sample the flow (grid-stride over the 2 B x 2H x 2W outputs, the arithmetic of resize_ac_kernel)
__global__ __launch_bounds__(256) void flow_warp_up2_tiled_kernel(const float* __restrict__ src, const float* __restrict__ flow, float* __restrict__ dst,
                                                                  float* __restrict__ flow_up, int B, int C, int H, int W, int ntiles) {
    fp16_saturate_on();
    __shared__ __attribute__((aligned(16))) float tile[3 * WB_PLANE];
    __shared__ int bx[4];
    const long long hw = (long long)H * W;
    if ((int)blockIdx.x < ntiles) {
        const int tid = threadIdx.x;
        int b, x, y;
        bool live;
        tile_pixel(H, W, b, x, y, live);
        const long long pix = (long long)(live ? y : 0) * W + (live ? x : 0);
        if (tid == 0) box_reset(bx);
        const float* fp = flow + (long long)b * 2 * hw + pix;
        const Taps t = make_taps((float)x + fp[0], (float)y + fp[hw], W, H);
        __syncthreads();
        box_add(bx, t, live, W, H);
        __syncthreads();
        const StagedBox sb = box_get(bx);
        for (int c0 = 0; c0 < C; c0 += 3) {
            const int nch = C - c0 < 3 ? C - c0 : 3;
            const float* sp = src + ((long long)b * C + c0) * hw;
            if (sb.ok) {
                if (c0) __syncthreads();
                box_stage(tile, sp, hw, W, nch, sb, tid);
                __syncthreads();
            }
            if (live)
                for (int c = 0; c < nch; ++c)
                    dst[((long long)b * C + c0 + c) * hw + pix] = sb.ok ? sample_tile(tile + c * WB_PLANE, t, sb) : sample_plane(sp + c * hw, t, W);
        }
        return;
    }
    const int Ho = 2 * H, Wo = 2 * W;
    const long long ohw = (long long)Ho * Wo;
    const long long nup = (long long)B * 2 * ohw;
    const float sh = (float)(H - 1) / (float)(Ho - 1), sw = (float)(W - 1) / (float)(Wo - 1);
    for (long long e = (long long)(blockIdx.x - ntiles) * blockDim.x + threadIdx.x; e < nup; e += (long long)(gridDim.x - ntiles) * blockDim.x) {
        const int p = (int)(e / ohw);                    // plane b * 2 + c
        const long long pix = e - (long long)p * ohw;
        const int oy = (int)(pix / Wo), ox = (int)(pix - (long long)oy * Wo);
        const float ry = sh * (float)oy, rx = sw * (float)ox;
        const int y0 = (int)ry, x0 = (int)rx;
        const int yp = (y0 < H - 1) ? 1 : 0, xp = (x0 < W - 1) ? 1 : 0;
        const float ly = ry - (float)y0, lx = rx - (float)x0;
        const float hy = 1.0f - ly, hx = 1.0f - lx;
        const float* s = flow + (long long)p * hw + (long long)y0 * W + x0;
        const float v = hy * (hx * s[0] + lx * s[xp]) + ly * (hx * s[yp * W] + lx * s[yp * W + xp]);
        flow_up[e] = v * 2.0f;
    }
}

__global__ __launch_bounds__(256) void warp_blend_tiled_kernel(const float* __restrict__ im0, const float* __restrict__ im1,
                                                               const float* __restrict__ motion, int motion_ld, long long motion_bstride,
                                                               float* __restrict__ i0w, float* __restrict__ i1w, float* __restrict__ it,
                                                               float* __restrict__ f0o, float* __restrict__ f1o,
                                                               float* __restrict__ m1o, float* __restrict__ m2o,
                                                               const float* __restrict__ orig0, const float* __restrict__ orig1,
                                                               float* __restrict__ pack15, int pack_ld, _Float16* __restrict__ pack_hi,
                                                               _Float16* __restrict__ pack_lo, long long pack_rows, int pack_c0, int B, int H,
                                                               int W) {
    fp16_saturate_on();
    __shared__ __attribute__((aligned(16))) float tile[2][3 * WB_PLANE];
    __shared__ int bx[2][4];
    const int tid = threadIdx.x;
    const long long hw = (long long)H * W;
    int b, x, y;
    bool live;
    tile_pixel(H, W, b, x, y, live);
    const long long pix = (long long)(live ? y : 0) * W + (live ? x : 0);
    if (tid < 2) box_reset(bx[tid]);
    const float* mp = motion + b * motion_bstride + pix * motion_ld;
    const float fx0 = mp[0], fy0 = mp[1], fx1 = mp[2], fy1 = mp[3];
    const float m1 = sigmoidf_(mp[4]);
    const float m2 = 1.0f - m1;
    const Taps t0 = make_taps((float)x + fx0, (float)y + fy0, W, H);
    const Taps t1 = make_taps((float)x + fx1, (float)y + fy1, W, H);
    __syncthreads();
    box_add(bx[0], t0, live, W, H);
    box_add(bx[1], t1, live, W, H);
    __syncthreads();
    const StagedBox s0 = box_get(bx[0]), s1 = box_get(bx[1]);
    if (s0.ok) box_stage(tile[0], im0 + (long long)b * 3 * hw, hw, W, 3, s0, tid);
    if (s1.ok) box_stage(tile[1], im1 + (long long)b * 3 * hw, hw, W, 3, s1, tid);
    __syncthreads();
    if (!live) return;
    float a[3], c[3], o[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const long long pl = ((long long)b * 3 + ch) * hw;
        a[ch] = s0.ok ? sample_tile(tile[0] + ch * WB_PLANE, t0, s0) : sample_plane(im0 + pl, t0, W);
        c[ch] = s1.ok ? sample_tile(tile[1] + ch * WB_PLANE, t1, s1) : sample_plane(im1 + pl, t1, W);
        o[ch] = m1 * a[ch] + m2 * c[ch];
        i0w[pl + pix] = a[ch];
        i1w[pl + pix] = c[ch];
        it[pl + pix] = o[ch];
    }
    if (f0o) {
        f0o[((long long)b * 2) * hw + pix] = fx0;
        f0o[((long long)b * 2 + 1) * hw + pix] = fy0;
        f1o[((long long)b * 2) * hw + pix] = fx1;
        f1o[((long long)b * 2 + 1) * hw + pix] = fy1;
    }
    if (m1o) {
        m1o[(long long)b * hw + pix] = m1;
        m2o[(long long)b * hw + pix] = m2;
    }
    if (pack15) {
        float* pp = pack15 + ((long long)b * hw + pix) * pack_ld;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const long long pl = ((long long)b * 3 + ch) * hw + pix;
            pp[ch] = orig0[pl];
            pp[3 + ch] = a[ch];
            pp[6 + ch] = orig1[pl];
            pp[9 + ch] = c[ch];
            pp[12 + ch] = o[ch];
        }
    }
    if (pack_hi) {
        float v16[16];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const long long pl = ((long long)b * 3 + ch) * hw + pix;
            v16[ch] = orig0[pl];
            v16[3 + ch] = a[ch];
            v16[6 + ch] = orig1[pl];
            v16[9 + ch] = c[ch];
            v16[12 + ch] = o[ch];
        }
        v16[15] = 0.f;
        const RowSink sink{nullptr, 0, pack_hi, pack_lo, pack_rows};
#pragma unroll
        for (int q = 0; q < 4; ++q)
            sink_store4(sink, (long long)b * hw + pix, pack_c0 + 4 * q, (f32x4){v16[4 * q], v16[4 * q + 1], v16[4 * q + 2], v16[4 * q + 3]});
    }
}

// ---------------------------------------------------------------------------------------
// align_corners=True bilinear resize (ATen upsample_bilinear2d arithmetic), planar
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resize_ac_kernel(const float* __restrict__ src, long long sb, long long sc, long long sy,
                                                        long long sx, float* __restrict__ dst, int B, int C,
                                                        int Hi, int Wi, int Ho, int Wo, float value_scale) {
    fp16_saturate_on();
    const float sh = (Ho > 1) ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f;
    const float sw = (Wo > 1) ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
    const long long ohw = (long long)Ho * Wo;
    const long long total = (long long)B * C * ohw;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int p = (int)(idx / ohw);
        const int pb = p / C, pc = p - pb * C;
        const long long pix = idx - (long long)p * ohw;
        const int oy = (int)(pix / Wo), ox = (int)(pix - (long long)oy * Wo);
        const float ry = sh * (float)oy, rx = sw * (float)ox;
        const int y0 = (int)ry, x0 = (int)rx;
        const int yp = (y0 < Hi - 1) ? 1 : 0, xp = (x0 < Wi - 1) ? 1 : 0;
        const float ly = ry - (float)y0, lx = rx - (float)x0;
        const float hy = 1.0f - ly, hx = 1.0f - lx;
        const float* s = src + pb * sb + pc * sc + y0 * sy + x0 * sx;
        const float v = hy * (hx * s[0] + lx * s[xp * sx]) + ly * (hx * s[yp * sy] + lx * s[yp * sy + xp * sx]);
        dst[idx] = v * value_scale;
    }
}

// The three x0.5 image-pyramid levels of both frames in ONE launch (network_base.py:444-448: F.interpolate(scale 0.5, bilinear,
// align_corners=True) applied level by level).  A thread owns one output element of one level and evaluates the levels below it on
// the fly with resize_ac_kernel's arithmetic, in its order -- bit-identical to three sequential resizes (level 3 costs 64 reads
// of the frame per element; it has 1 / 64 of the frame's pixels).  Four dependent 7-us launches become one.
template <int L>
__device__ __forceinline__ float pyramid_value(const float* __restrict__ plane, int H, int W, int oy, int ox) {
    if constexpr (L == 0) {
        return plane[(long long)oy * W + ox];
    } else {
        const int Hi = H >> (L - 1), Wi = W >> (L - 1), Ho = H >> L, Wo = W >> L;
        const float sh = (Ho > 1) ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f;
        const float sw = (Wo > 1) ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
        const float ry = sh * (float)oy, rx = sw * (float)ox;
        const int y0 = (int)ry, x0 = (int)rx;
        const int yp = (y0 < Hi - 1) ? 1 : 0, xp = (x0 < Wi - 1) ? 1 : 0;
        const float ly = ry - (float)y0, lx = rx - (float)x0;
        const float hy = 1.0f - ly, hx = 1.0f - lx;
        const float v00 = pyramid_value<L - 1>(plane, H, W, y0, x0), v01 = pyramid_value<L - 1>(plane, H, W, y0, x0 + xp);
        const float v10 = pyramid_value<L - 1>(plane, H, W, y0 + yp, x0), v11 = pyramid_value<L - 1>(plane, H, W, y0 + yp, x0 + xp);
        const float v = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
        return v * 1.0f;
    }
}

__global__ __launch_bounds__(256) void image_pyramid_kernel(const float* __restrict__ im0, const float* __restrict__ im1, float* __restrict__ l1,
                                                            float* __restrict__ l2, float* __restrict__ l3, float* __restrict__ pack, int B, int H, int W) {
    fp16_saturate_on();
    const long long n1 = (long long)(H >> 1) * (W >> 1), n2 = (long long)(H >> 2) * (W >> 2), n3 = (long long)(H >> 3) * (W >> 3);
    const long long planes = 2ll * B * 3;
    const long long t1 = planes * n1, t2 = planes * n2, t3 = planes * n3;
    const long long hw = (long long)H * W, tp = pack ? 2ll * B * hw : 0;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < t1 + t2 + t3 + tp; idx += (long long)gridDim.x * blockDim.x) {
        if (idx >= t1 + t2 + t3) {
            // torch.cat([im0, im1], 0) as NHWC4 (the encoder's input; pack_frames_kernel's work in the same launch)
            const long long e = idx - (t1 + t2 + t3);
            const int f = (int)(e / hw);
            const long long pix = e - (long long)f * hw;
            const float* s = (f < B ? im0 + (long long)f * 3 * hw : im1 + (long long)(f - B) * 3 * hw) + pix;
            *reinterpret_cast<f32x4*>(pack + e * 4) = (f32x4){s[0], s[hw], s[2 * hw], 0.f};
            continue;
        }
        const int lvl = idx < t1 ? 1 : (idx < t1 + t2 ? 2 : 3);
        const long long e = idx - (lvl == 1 ? 0 : (lvl == 2 ? t1 : t1 + t2));
        const long long n = lvl == 1 ? n1 : (lvl == 2 ? n2 : n3);
        const int pl = (int)(e / n);                                   // stacked plane: [im0 batch x 3, im1 batch x 3]
        const long long pix = e - (long long)pl * n;
        const int Wo = W >> lvl;
        const int oy = (int)(pix / Wo), ox = (int)(pix - (long long)oy * Wo);
        const float* src = (pl < 3 * B ? im0 + (long long)pl * H * W : im1 + (long long)(pl - 3 * B) * H * W);
        if (lvl == 1) l1[e] = pyramid_value<1>(src, H, W, oy, ox);
        else if (lvl == 2) l2[e] = pyramid_value<2>(src, H, W, oy, ox);
        else l3[e] = pyramid_value<3>(src, H, W, oy, ox);
    }
}

__global__ __launch_bounds__(256) void pack_frames_kernel(const float* __restrict__ im0, const float* __restrict__ im1,
                                                          float* __restrict__ dst, int B, int H, int W) {
    fp16_saturate_on();
    const long long hw = (long long)H * W;
    const long long total = 2ll * B * hw;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int f = (int)(idx / hw);           // stacked frame index: [im0 batch..., im1 batch...]
        const long long pix = idx - (long long)f * hw;
        const float* s = (f < B ? im0 + (long long)f * 3 * hw : im1 + (long long)(f - B) * 3 * hw) + pix;
        *reinterpret_cast<f32x4*>(dst + idx * 4) = (f32x4){s[0], s[hw], s[2 * hw], 0.f};
    }
}

// ---------------------------------------------------------------------------------------
// Host-boundary frames (demo_2x.py:64-85 inference_2frame, benchmark/utils.py:57-80 InputPadder):
//   uint8 HWC (BGR or RGB) -> fp32 planar RGB in [0,1] (x / 255, a true fp32 division as torch does), centred replicate padding
//   fp32 planar -> unpad -> np.round(x * 255) (round half to even) -> uint8 HWC
// Integer / byte work: bit-exact against the numpy path by construction.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void frame_u8_to_f32_kernel(const unsigned char* __restrict__ src, int H, int W, int bgr,
                                                              float* __restrict__ dst, int Hp, int Wp, int pad_top, int pad_left) {
    fp16_saturate_on();
    const long long total = (long long)Hp * Wp;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(idx / Wp), x = (int)(idx - (long long)y * Wp);
        int sy = y - pad_top, sx = x - pad_left;                 // replicate: clamp to the source frame
        sy = sy < 0 ? 0 : (sy >= H ? H - 1 : sy);
        sx = sx < 0 ? 0 : (sx >= W ? W - 1 : sx);
        const unsigned char* p = src + ((long long)sy * W + sx) * 3;
        const float c0 = (float)p[0] / 255.0f, c1 = (float)p[1] / 255.0f, c2 = (float)p[2] / 255.0f;
        dst[idx] = bgr ? c2 : c0;
        dst[total + idx] = c1;
        dst[2 * total + idx] = bgr ? c0 : c2;
    }
}

__global__ __launch_bounds__(256) void frame_f32_to_u8_kernel(const float* __restrict__ src, int Hp, int Wp, int pad_top, int pad_left,
                                                              unsigned char* __restrict__ dst, int H, int W, int bgr) {
    fp16_saturate_on();
    const long long total = (long long)H * W, plane = (long long)Hp * Wp;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(idx / W), x = (int)(idx - (long long)y * W);
        const float* p = src + (long long)(y + pad_top) * Wp + (x + pad_left);
        int v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int q = __float2int_rn(p[c * plane] * 255.0f);        // rint: half to even, as np.round
            v[c] = q < 0 ? 0 : (q > 255 ? 255 : q);
        }
        unsigned char* o = dst + idx * 3;
        o[0] = (unsigned char)(bgr ? v[2] : v[0]);
        o[1] = (unsigned char)v[1];
        o[2] = (unsigned char)(bgr ? v[0] : v[2]);
    }
}

__global__ __launch_bounds__(256) void final_residual_kernel(const float* __restrict__ it, const float* __restrict__ r, int r_ld,
                                                             float* __restrict__ it_sum, float* __restrict__ it_clamped,
                                                             int B, int H, int W) {
    fp16_saturate_on();
    const long long hw = (long long)H * W;
    const long long total = (long long)B * hw;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(idx / hw);
        const long long pix = idx - (long long)b * hw;
        const float* rp = r + idx * r_ld;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const long long o = ((long long)b * 3 + ch) * hw + pix;
            const float v = it[o] + (2.0f * sigmoidf_(rp[ch]) - 1.0f);
            it_sum[o] = v;
            it_clamped[o] = fminf(fmaxf(v, 0.0f), 1.0f);
        }
    }
}

// refine_head.1 (nn.Conv2d(C, 3, 3, padding 1) + PReLU) -> 2 * sigmoid - 1 -> += I_t -> clamp (network_base.py:257-260, 429, 532-533) from
// the per-pixel tap contributions T[(tap * 3 + o)][pixel] = sum_c w[o][c][tap] * r1[pixel][c] that atmvfi_conv3x3_planes_readout left
// (planar fp32): out[o][p] = bias[o] + sum over taps of T[tap * 3 + o][p + offset(tap)], taps outside the image contributing nothing
// (the zero padding of the convolution).  One thread per pixel; every load is coalesced.
__global__ __launch_bounds__(256) void refine_tail_kernel(const float* __restrict__ T, long long plane, const float* __restrict__ bias,
                                                          const float* __restrict__ slope, const float* __restrict__ it,
                                                          float* __restrict__ it_sum, float* __restrict__ it_clamped, int B, int H, int W) {
    fp16_saturate_on();
    const long long hw = (long long)H * W;
    const long long total = (long long)B * hw;
    const float b0 = bias ? bias[0] : 0.f, b1 = bias ? bias[1] : 0.f, b2 = bias ? bias[2] : 0.f;
    const float s0 = slope ? slope[0] : 1.f, s1 = slope ? slope[1] : 1.f, s2 = slope ? slope[2] : 1.f;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(idx / hw);
        const long long pix = idx - (long long)b * hw;
        const int y = (int)(pix / W), x = (int)(pix - (long long)y * W);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int yy = y + ky - 1, xx = x + kx - 1;
                if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
                    const float* tp = T + (long long)((ky * 3 + kx) * 3) * plane + idx + (long long)(ky - 1) * W + (kx - 1);
                    a0 += tp[0];
                    a1 += tp[plane];
                    a2 += tp[2 * plane];
                }
            }
        float r[3] = {a0 + b0, a1 + b1, a2 + b2};
        r[0] = r[0] > 0.f ? r[0] : s0 * r[0];
        r[1] = r[1] > 0.f ? r[1] : s1 * r[1];
        r[2] = r[2] > 0.f ? r[2] : s2 * r[2];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const long long o = ((long long)b * 3 + ch) * hw + pix;
            const float v = it[o] + (2.0f * sigmoidf_(r[ch]) - 1.0f);
            it_sum[o] = v;
            it_clamped[o] = fminf(fmaxf(v, 0.0f), 1.0f);
        }
    }
}

// mean |a - b| per sample in two passes with a FIXED summation order (run-to-run bit-identical: the ensemble's pick compares these means,
// and a launch plan's self-check replays the forward): pass 1 = one partial sum per block (wave shuffle tree, then the four waves in
// order), pass 2 = one block per sample adds the partial sums, thread by thread in index order, then the same tree.
__global__ __launch_bounds__(256) void l1_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ part, long long per_sample) {
    __shared__ float red[4];
    const int s = blockIdx.y;
    const float* pa = a + (long long)s * per_sample;
    const float* pb = b + (long long)s * per_sample;
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per_sample; i += (long long)gridDim.x * blockDim.x)
        acc += fabsf(pa[i] - pb[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[(long long)s * gridDim.x + blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}
__global__ __launch_bounds__(256) void l1_final_kernel(const float* __restrict__ part, int nparts, float* __restrict__ out, long long per_sample) {
    __shared__ float red[4];
    const int s = blockIdx.x;
    float acc = 0.f;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) acc += part[(long long)s * nparts + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[s] = (((red[0] + red[1]) + red[2]) + red[3]) / (float)per_sample;
}

// multiscale_global_motion_ensemble's per-sample pick (network_base.py:591-603): the candidate whose alignment loss is the minimum, the
// FIRST one on ties (the reference's if / elif chain over levels 0, 1, 2), copied into the output flows.  Candidates are already at the
// level-0 flow resolution (the x2 / x4 up-sampling of levels 1 / 2 is the candidates' resize).  The minimum is Python's min(l0, l1, l2)
// -- start from l0, replace by a later value only if that one compares LESS -- and the pick its `==` chain: with a NaN loss every
// comparison is false, so a NaN l0 makes the chain fall through to level 2, exactly as the reference does (fminf would skip the NaN).
__global__ __launch_bounds__(256) void ensemble_select_kernel(const float* __restrict__ l0, const float* __restrict__ l1, const float* __restrict__ l2,
                                                              const float* __restrict__ a0, const float* __restrict__ b0,
                                                              const float* __restrict__ a1, const float* __restrict__ b1,
                                                              const float* __restrict__ a2, const float* __restrict__ b2,
                                                              float* __restrict__ out0, float* __restrict__ out1, int B, long long per_sample) {
    fp16_saturate_on();
    const long long total = (long long)B * per_sample;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int s = (int)(idx / per_sample);
        const float x0 = l0[s], x1 = l1[s], x2 = l2[s];
        float mn = x0;
        if (x1 < mn) mn = x1;
        if (x2 < mn) mn = x2;
        const int pick = x0 == mn ? 0 : (x1 == mn ? 1 : 2);
        out0[idx] = pick == 0 ? a0[idx] : (pick == 1 ? a1[idx] : a2[idx]);
        out1[idx] = pick == 0 ? b0[idx] : (pick == 1 ? b1[idx] : b2[idx]);
    }
}

inline unsigned grid_for(long long total) {
    long long b = (total + 255) / 256;
    if (b > 8192) b = 8192;   // 256 CUs x 32 blocks, grid-stride the rest
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" int atmvfi_layernorm(const float* in, int in_ld, int64_t in_gstride, int in_rpg, const int32_t* src_row_map,
                                 float* out, int out_ld, const float* gamma, const float* beta, int64_t rows, int C,
                                 void* out_hi, void* out_lo, int plane_ld, void* stream) {
    ATMVFI_REQUIRE(in && gamma && beta, ATMVFI_EINVAL, "layernorm: null pointer");
    ATMVFI_REQUIRE(rows > 0 && C > 0 && C % 4 == 0, ATMVFI_EINVAL, "layernorm: C %d must be a positive multiple of 4", C);
    ATMVFI_REQUIRE(in_ld % 4 == 0 && in_ld >= C && in_gstride % 4 == 0, ATMVFI_EALIGN,
                   "layernorm: leading dimensions must be multiples of 4 and >= C");
    ATMVFI_REQUIRE(sink_ok(out, out_ld, C, out_hi, out_lo, plane_ld, rows), ATMVFI_EALIGN,
                   "layernorm: output needs fp32 rows (ld %% 4 == 0, >= C) and/or both fp16 planes (plane rows >= rows), 16-byte aligned");
    ATMVFI_REQUIRE(atmvfi::aligned16(in) && atmvfi::aligned16(gamma) && atmvfi::aligned16(beta),
                   ATMVFI_EALIGN, "layernorm: pointers must be 16-byte aligned");
    const unsigned blocks = (unsigned)((rows + 3) / 4);
    const RowSink sink{out, out_ld, (_Float16*)out_hi, (_Float16*)out_lo, plane_ld};
    hipLaunchKernelGGL(layernorm_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, in_ld, (long long)in_gstride,
                       in_rpg, src_row_map, sink, gamma, beta, (long long)rows, C);
    return atmvfi::check_launch("layernorm");
}

extern "C" int atmvfi_dwconv3x3_gelu(const float* in, int in_ld, float* out, int out_ld, const float* weight9,
                                      const float* bias, int N, int H, int W, int C, void* out_hi, void* out_lo,
                                      int plane_ld, void* stream) {
    ATMVFI_REQUIRE(in && weight9 && bias, ATMVFI_EINVAL, "dwconv3x3_gelu: null pointer");
    ATMVFI_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, ATMVFI_EINVAL, "dwconv3x3_gelu: bad shape");
    ATMVFI_REQUIRE(in_ld % 4 == 0 && in_ld >= C, ATMVFI_EALIGN, "dwconv3x3_gelu: bad ld");
    ATMVFI_REQUIRE(sink_ok(out, out_ld, C, out_hi, out_lo, plane_ld, (long long)N * H * W), ATMVFI_EALIGN,
                   "dwconv3x3_gelu: output needs fp32 rows (ld %% 4 == 0, >= C) and/or both fp16 planes (plane rows >= N*H*W), 16-byte aligned");
    const RowSink sink{out, out_ld, (_Float16*)out_hi, (_Float16*)out_lo, plane_ld};
    ATMVFI_REQUIRE(atmvfi::aligned16(in) && atmvfi::aligned16(out) && atmvfi::aligned16(weight9) && atmvfi::aligned16(bias),
                   ATMVFI_EALIGN, "dwconv3x3_gelu: pointers must be 16-byte aligned");
#if defined(ATMVFI_ABLATE)
    static const bool rows_only = getenv("ATMVFI_DWCONV_ROWS") != nullptr;     // diagnostic builds only (make ablate): same-box A/B against the sliding-window kernel, tools/dw_ab.sh
#else
    constexpr bool rows_only = false;
#endif
    if (!out && C % 64 == 0 && H >= 8 && !rows_only) {
        // plane sink only: the LDS-DMA kernel (one wave = 8 x-positions x 32 channels x RS rows; C % 64 as for the sliding-window kernel,
        // whose fused multiply-adds it repeats -- the per-pixel kernel below rounds products and sums separately).  RS: the tallest strip that divides H
        // (1088 / 8 = 136 = 8 x 17), else 16 (or 8) with the last strip overlapping the one before
        const int xgroups = (W + 7) / 8, cblocks = C / 32;
        int RS = H % 17 == 0 ? 17 : H % 16 == 0 ? 16 : H % 8 == 0 ? 8 : H >= 16 ? 16 : 8;
        // (small maps: 8-row strips when the taller ones leave fewer than two waves per SIMD)
        if ((long long)N * ((H + RS - 1) / RS) * xgroups * cblocks < 8ll * atmvfi::cu_count()) RS = 8;
        const int strips = (H + RS - 1) / RS;
        const long long units = (long long)N * strips * xgroups * cblocks;
        const long long blocks = (units + 3) / 4;
        ATMVFI_REQUIRE(units < (1ll << 31), ATMVFI_EINVAL, "dwconv3x3_gelu: grid too large");
        const size_t lds = 4 * DW_RING * DW_SLOT;
        auto go = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, in, in_ld, (_Float16*)out_hi,
                               (_Float16*)out_lo, (long long)plane_ld, weight9, bias, N, H, W, C, xgroups, cblocks, strips);
        };
        if (RS == 17) go(dwconv_gelu_dma_kernel<17>);
        else if (RS == 16) go(dwconv_gelu_dma_kernel<16>);
        else go(dwconv_gelu_dma_kernel<8>);
        return atmvfi::check_launch("dwconv3x3_gelu");
    }
    if (C % 64 == 0) {     // sliding-window kernel: 16 x-positions x 64 channels per block, strips of 8 or 16 rows
        // a strip re-reads its two neighbour rows: 10 / 8 of the input with 8-row strips, 18 / 16 with 16-row ones; the taller strip
        // when it still leaves a few blocks per CU
        const int xblocks = (W + 15) / 16, cblocks = C / 64;
        const bool tall = (long long)N * ((H + 15) / 16) * xblocks * cblocks >= 8ll * atmvfi::cu_count();
        const int RS = tall ? 16 : 8;
        const int strips = (H + RS - 1) / RS;
        const long long blocks = (long long)N * strips * xblocks * cblocks;
        ATMVFI_REQUIRE(blocks < (1ll << 31), ATMVFI_EINVAL, "dwconv3x3_gelu: grid too large");
        if (tall)
            hipLaunchKernelGGL(dwconv_gelu_rows_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, in_ld, sink,
                               weight9, bias, N, H, W, C, xblocks, cblocks, strips);
        else
            hipLaunchKernelGGL(dwconv_gelu_rows_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, in_ld, sink,
                               weight9, bias, N, H, W, C, xblocks, cblocks, strips);
        return atmvfi::check_launch("dwconv3x3_gelu");
    }
    const long long total = (long long)N * H * W * (C / 4);
    hipLaunchKernelGGL(dwconv_gelu_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, in_ld, sink,
                       weight9, bias, N, H, W, C);
    return atmvfi::check_launch("dwconv3x3_gelu");
}

extern "C" int atmvfi_pack_dw_weight(const float* src, float* dst, int C, void* stream) {
    ATMVFI_REQUIRE(src && dst && C > 0, ATMVFI_EINVAL, "pack_dw_weight: bad arguments");
    hipLaunchKernelGGL(pack_dw_kernel, dim3((9 * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, dst, C);
    return atmvfi::check_launch("pack_dw_weight");
}

extern "C" int atmvfi_flow_warp(const float* src, const float* flow, int64_t flow_bstride, int flow_pstride,
                                 int flow_cstride, float* dst, int B, int C, int H, int W, void* stream) {
    ATMVFI_REQUIRE(src && flow && dst, ATMVFI_EINVAL, "flow_warp: null pointer");
    ATMVFI_REQUIRE(B > 0 && C > 0 && H > 1 && W > 1, ATMVFI_EINVAL, "flow_warp: bad shape (H, W must be > 1)");
    hipLaunchKernelGGL(flow_warp_planar_kernel, dim3(grid_for((long long)B * H * W)), dim3(256), 0, (hipStream_t)stream, src,
                       flow, (long long)flow_bstride, flow_pstride, flow_cstride, dst, B, C, H, W);
    return atmvfi::check_launch("flow_warp");
}

extern "C" int atmvfi_flow_warp_ex(const float* src, const float* flow, float* dst, uint8_t* mask, int B, int C, int H, int W,
                                    int padding_mode, void* stream) {
    ATMVFI_REQUIRE(src && flow && dst, ATMVFI_EINVAL, "flow_warp_ex: null pointer");
    ATMVFI_REQUIRE(B > 0 && C > 0 && H > 1 && W > 1, ATMVFI_EINVAL, "flow_warp_ex: bad shape (H, W must be > 1)");
    ATMVFI_REQUIRE(padding_mode >= 0 && padding_mode <= 2, ATMVFI_EINVAL, "flow_warp_ex: padding_mode must be 0 (zeros), 1 (border) or 2 (reflection), got %d", padding_mode);
    hipLaunchKernelGGL(flow_warp_ex_kernel, dim3(grid_for((long long)B * H * W)), dim3(256), 0, (hipStream_t)stream, src, flow, dst,
                       (unsigned char*)mask, B, C, H, W, padding_mode);
    return atmvfi::check_launch("flow_warp_ex");
}

extern "C" int atmvfi_flow_warp_tiled(const float* src, const float* flow, int64_t flow_bstride, int flow_pstride,
                                       int flow_cstride, float* dst, int B, int C, int H, int W, void* stream) {
    ATMVFI_REQUIRE(src && flow && dst, ATMVFI_EINVAL, "flow_warp_tiled: null pointer");
    ATMVFI_REQUIRE(B > 0 && C > 0 && H > 1 && W > 1, ATMVFI_EINVAL, "flow_warp_tiled: bad shape (H, W must be > 1)");
    ATMVFI_REQUIRE(W % 4 == 0 && atmvfi::aligned16(src), ATMVFI_EINVAL, "flow_warp_tiled: W must be a multiple of 4 and src 16-byte aligned (got W = %d)", W);
    const long long tiles = (long long)B * ((H + WT_H - 1) / WT_H) * ((W + WT_W - 1) / WT_W);
    ATMVFI_REQUIRE(tiles < (1ll << 31), ATMVFI_EINVAL, "flow_warp_tiled: grid too large");
    hipLaunchKernelGGL(flow_warp_tiled_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, src, flow, (long long)flow_bstride,
                       flow_pstride, flow_cstride, dst, B, C, H, W);
    return atmvfi::check_launch("flow_warp_tiled");
}

extern "C" int atmvfi_flow_warp_up2(const float* src, const float* flow, float* dst, float* flow_up, int B, int C, int H, int W, void* stream) {
    ATMVFI_REQUIRE(src && flow && dst && flow_up, ATMVFI_EINVAL, "flow_warp_up2: null pointer");
    ATMVFI_REQUIRE(B > 0 && C > 0 && H > 1 && W > 1, ATMVFI_EINVAL, "flow_warp_up2: bad shape (H, W must be > 1)");
    hipLaunchKernelGGL(flow_warp_up2_kernel, dim3(grid_for((long long)B * H * W * 9)), dim3(256), 0, (hipStream_t)stream, src, flow, dst, flow_up,
                       B, C, H, W);
    return atmvfi::check_launch("flow_warp_up2");
}

extern "C" int atmvfi_flow_warp_up2_tiled(const float* src, const float* flow, float* dst, float* flow_up, int B, int C, int H, int W, void* stream) {
    ATMVFI_REQUIRE(src && flow && dst && flow_up, ATMVFI_EINVAL, "flow_warp_up2_tiled: null pointer");
    ATMVFI_REQUIRE(B > 0 && C > 0 && H > 1 && W > 1, ATMVFI_EINVAL, "flow_warp_up2_tiled: bad shape (H, W must be > 1)");
    ATMVFI_REQUIRE(W % 4 == 0 && atmvfi::aligned16(src), ATMVFI_EINVAL, "flow_warp_up2_tiled: W must be a multiple of 4 and src 16-byte aligned (got W = %d)", W);
    const long long tiles = (long long)B * ((H + WT_H - 1) / WT_H) * ((W + WT_W - 1) / WT_W);
    const long long upblocks = grid_for((long long)B * H * W * 8);
    ATMVFI_REQUIRE(tiles + upblocks < (1ll << 31), ATMVFI_EINVAL, "flow_warp_up2_tiled: grid too large");
    hipLaunchKernelGGL(flow_warp_up2_tiled_kernel, dim3((unsigned)(tiles + upblocks)), dim3(256), 0, (hipStream_t)stream, src, flow, dst, flow_up,
                       B, C, H, W, (int)tiles);
    return atmvfi::check_launch("flow_warp_up2_tiled");
}

extern "C" int atmvfi_flow_warp_nhwc(const float* src, int src_ld, int64_t src_bstride, const float* flow,
                                      int64_t flow_bstride, int flow_pstride, int flow_cstride, float* dst, int dst_ld,
                                      int64_t dst_bstride, int B, int C, int H, int W, void* stream) {
    ATMVFI_REQUIRE(src && flow && dst, ATMVFI_EINVAL, "flow_warp_nhwc: null pointer");
    ATMVFI_REQUIRE(B > 0 && C > 0 && C % 4 == 0 && H > 1 && W > 1, ATMVFI_EINVAL, "flow_warp_nhwc: bad shape");
    ATMVFI_REQUIRE(src_ld % 4 == 0 && dst_ld % 4 == 0 && src_bstride % 4 == 0 && dst_bstride % 4 == 0 &&
                       atmvfi::aligned16(src) && atmvfi::aligned16(dst),
                   ATMVFI_EALIGN, "flow_warp_nhwc: views must be 16-byte aligned");
    hipLaunchKernelGGL(flow_warp_nhwc_kernel, dim3(grid_for((long long)B * H * W * (C / 4))), dim3(256), 0,
                       (hipStream_t)stream, src, src_ld, (long long)src_bstride, flow, (long long)flow_bstride, flow_pstride,
                       flow_cstride, dst, dst_ld, (long long)dst_bstride, B, C, H, W);
    return atmvfi::check_launch("flow_warp_nhwc");
}

extern "C" int atmvfi_warp_blend_planes(const float* im0, const float* im1, const float* motion, int motion_ld,
                                         int64_t motion_bstride, float* i0w, float* i1w, float* it, float* flow0_out,
                                         float* flow1_out, float* mask1_out, float* mask2_out, const float* orig0,
                                         const float* orig1, float* pack15, int pack_ld, void* pack_hi, void* pack_lo, int64_t pack_rows,
                                         int pack_c0, int B, int H, int W, void* stream) {
    ATMVFI_REQUIRE(im0 && im1 && motion && i0w && i1w && it, ATMVFI_EINVAL, "warp_blend: null pointer");
    ATMVFI_REQUIRE(B > 0 && H > 1 && W > 1 && motion_ld >= 5, ATMVFI_EINVAL, "warp_blend: bad shape");
    ATMVFI_REQUIRE((flow0_out == nullptr) == (flow1_out == nullptr) && (mask1_out == nullptr) == (mask2_out == nullptr),
                   ATMVFI_EINVAL, "warp_blend: flow/mask outputs come in pairs");
    if (pack15) ATMVFI_REQUIRE(orig0 && orig1 && pack_ld >= 15, ATMVFI_EINVAL, "warp_blend: pack15 needs orig0/orig1 and ld >= 15");
    ATMVFI_REQUIRE((pack_hi == nullptr) == (pack_lo == nullptr), ATMVFI_EINVAL, "warp_blend: the plane sink needs both planes");
    if (pack_hi)
        ATMVFI_REQUIRE(orig0 && orig1 && pack_rows >= (int64_t)B * H * W && pack_c0 >= 0 && pack_c0 % 4 == 0 && atmvfi::aligned16(pack_hi) &&
                           atmvfi::aligned16(pack_lo), ATMVFI_EINVAL,
                       "warp_blend: plane sink needs orig0/orig1, rows >= B*H*W, a channel offset that is a multiple of 4, aligned planes");
    hipLaunchKernelGGL(warp_blend_kernel, dim3(grid_for((long long)B * H * W)), dim3(256), 0, (hipStream_t)stream, im0, im1,
                       motion, motion_ld, (long long)motion_bstride, i0w, i1w, it, flow0_out, flow1_out, mask1_out, mask2_out,
                       orig0, orig1, pack15, pack_ld, (_Float16*)pack_hi, (_Float16*)pack_lo, (long long)pack_rows, pack_c0, B, H, W);
    return atmvfi::check_launch("warp_blend");
}
extern "C" int atmvfi_warp_blend_tiled(const float* im0, const float* im1, const float* motion, int motion_ld,
                                        int64_t motion_bstride, float* i0w, float* i1w, float* it, float* flow0_out,
                                        float* flow1_out, float* mask1_out, float* mask2_out, const float* orig0,
                                        const float* orig1, float* pack15, int pack_ld, void* pack_hi, void* pack_lo, int64_t pack_rows,
                                        int pack_c0, int B, int H, int W, void* stream) {
    ATMVFI_REQUIRE(im0 && im1 && motion && i0w && i1w && it, ATMVFI_EINVAL, "warp_blend_tiled: null pointer");
    ATMVFI_REQUIRE(B > 0 && H > 1 && W > 1 && motion_ld >= 5, ATMVFI_EINVAL, "warp_blend_tiled: bad shape");
    ATMVFI_REQUIRE(W % 4 == 0 && atmvfi::aligned16(im0) && atmvfi::aligned16(im1), ATMVFI_EINVAL,
                   "warp_blend_tiled: W must be a multiple of 4 and the images 16-byte aligned (got W = %d)", W);
    ATMVFI_REQUIRE((flow0_out == nullptr) == (flow1_out == nullptr) && (mask1_out == nullptr) == (mask2_out == nullptr),
                   ATMVFI_EINVAL, "warp_blend_tiled: flow/mask outputs come in pairs");
    if (pack15) ATMVFI_REQUIRE(orig0 && orig1 && pack_ld >= 15, ATMVFI_EINVAL, "warp_blend_tiled: pack15 needs orig0/orig1 and ld >= 15");
    ATMVFI_REQUIRE((pack_hi == nullptr) == (pack_lo == nullptr), ATMVFI_EINVAL, "warp_blend_tiled: the plane sink needs both planes");
    if (pack_hi)
        ATMVFI_REQUIRE(orig0 && orig1 && pack_rows >= (int64_t)B * H * W && pack_c0 >= 0 && pack_c0 % 4 == 0 && atmvfi::aligned16(pack_hi) &&
                           atmvfi::aligned16(pack_lo), ATMVFI_EINVAL,
                       "warp_blend_tiled: plane sink needs orig0/orig1, rows >= B*H*W, a channel offset that is a multiple of 4, aligned planes");
    const long long tiles = (long long)B * ((H + WT_H - 1) / WT_H) * ((W + WT_W - 1) / WT_W);
    ATMVFI_REQUIRE(tiles < (1ll << 31), ATMVFI_EINVAL, "warp_blend_tiled: grid too large");
    hipLaunchKernelGGL(warp_blend_tiled_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, im0, im1,
                       motion, motion_ld, (long long)motion_bstride, i0w, i1w, it, flow0_out, flow1_out, mask1_out, mask2_out,
                       orig0, orig1, pack15, pack_ld, (_Float16*)pack_hi, (_Float16*)pack_lo, (long long)pack_rows, pack_c0, B, H, W);
    return atmvfi::check_launch("warp_blend_tiled");
}
extern "C" int atmvfi_warp_blend(const float* im0, const float* im1, const float* motion, int motion_ld,
                                  int64_t motion_bstride, float* i0w, float* i1w, float* it, float* flow0_out,
                                  float* flow1_out, float* mask1_out, float* mask2_out, const float* orig0,
                                  const float* orig1, float* pack15, int pack_ld, int B, int H, int W, void* stream) {
    return atmvfi_warp_blend_planes(im0, im1, motion, motion_ld, motion_bstride, i0w, i1w, it, flow0_out, flow1_out, mask1_out, mask2_out,
                                    orig0, orig1, pack15, pack_ld, nullptr, nullptr, 0, 0, B, H, W, stream);
}

extern "C" int atmvfi_resize_bilinear_ac(const float* src, int64_t src_bstride, int64_t src_cstride, int64_t src_ystride,
                                          int64_t src_xstride, float* dst, int B, int C, int Hi, int Wi, int Ho, int Wo,
                                          float value_scale, void* stream) {
    ATMVFI_REQUIRE(src && dst, ATMVFI_EINVAL, "resize_bilinear_ac: null pointer");
    ATMVFI_REQUIRE(B > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, ATMVFI_EINVAL, "resize_bilinear_ac: bad shape");
    hipLaunchKernelGGL(resize_ac_kernel, dim3(grid_for((long long)B * C * Ho * Wo)), dim3(256), 0, (hipStream_t)stream, src,
                       (long long)src_bstride, (long long)src_cstride, (long long)src_ystride, (long long)src_xstride, dst,
                       B, C, Hi, Wi, Ho, Wo, value_scale);
    return atmvfi::check_launch("resize_bilinear_ac");
}

extern "C" int atmvfi_image_pyramid_pack(const float* im0, const float* im1, float* l1, float* l2, float* l3, float* pack, int B, int H, int W,
                                         void* stream) {
    ATMVFI_REQUIRE(im0 && im1 && l1 && l2 && l3, ATMVFI_EINVAL, "image_pyramid: null pointer");
    ATMVFI_REQUIRE(B > 0 && H >= 8 && W >= 8 && H % 8 == 0 && W % 8 == 0, ATMVFI_EINVAL, "image_pyramid: H, W must be multiples of 8 (got %dx%d)", H, W);
    ATMVFI_REQUIRE(!pack || atmvfi::aligned16(pack), ATMVFI_EALIGN, "image_pyramid: pack must be 16-byte aligned");
    const long long total = 2ll * B * 3 * ((long long)(H >> 1) * (W >> 1) + (long long)(H >> 2) * (W >> 2) + (long long)(H >> 3) * (W >> 3)) +
                            (pack ? 2ll * B * H * W : 0);
    hipLaunchKernelGGL(image_pyramid_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, im0, im1, l1, l2, l3, pack, B, H, W);
    return atmvfi::check_launch("image_pyramid");
}

extern "C" int atmvfi_image_pyramid(const float* im0, const float* im1, float* l1, float* l2, float* l3, int B, int H, int W, void* stream) {
    return atmvfi_image_pyramid_pack(im0, im1, l1, l2, l3, nullptr, B, H, W, stream);
}

extern "C" int atmvfi_pack_frames(const float* im0, const float* im1, float* dst, int B, int H, int W, void* stream) {
    ATMVFI_REQUIRE(im0 && im1 && dst && B > 0 && H > 0 && W > 0, ATMVFI_EINVAL, "pack_frames: bad arguments");
    ATMVFI_REQUIRE(atmvfi::aligned16(dst), ATMVFI_EALIGN, "pack_frames: dst must be 16-byte aligned");
    hipLaunchKernelGGL(pack_frames_kernel, dim3(grid_for(2ll * B * H * W)), dim3(256), 0, (hipStream_t)stream, im0, im1, dst,
                       B, H, W);
    return atmvfi::check_launch("pack_frames");
}

extern "C" int atmvfi_frame_u8_to_f32(const void* src, int H, int W, int bgr, float* dst, int Hp, int Wp, int pad_top, int pad_left,
                                       void* stream) {
    ATMVFI_REQUIRE(src && dst && H > 0 && W > 0 && Hp >= H && Wp >= W && pad_top >= 0 && pad_left >= 0 && pad_top + H <= Hp &&
                       pad_left + W <= Wp,
                   ATMVFI_EINVAL, "frame_u8_to_f32: bad geometry (H %d W %d -> Hp %d Wp %d, pad %d %d)", H, W, Hp, Wp, pad_top, pad_left);
    hipLaunchKernelGGL(frame_u8_to_f32_kernel, dim3(grid_for((long long)Hp * Wp)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned char*)src, H, W, bgr, dst, Hp, Wp, pad_top, pad_left);
    return atmvfi::check_launch("frame_u8_to_f32");
}

extern "C" int atmvfi_frame_f32_to_u8(const float* src, int Hp, int Wp, int pad_top, int pad_left, void* dst, int H, int W, int bgr,
                                       void* stream) {
    ATMVFI_REQUIRE(src && dst && H > 0 && W > 0 && Hp >= H && Wp >= W && pad_top >= 0 && pad_left >= 0 && pad_top + H <= Hp &&
                       pad_left + W <= Wp,
                   ATMVFI_EINVAL, "frame_f32_to_u8: bad geometry (Hp %d Wp %d -> H %d W %d, pad %d %d)", Hp, Wp, H, W, pad_top, pad_left);
    hipLaunchKernelGGL(frame_f32_to_u8_kernel, dim3(grid_for((long long)H * W)), dim3(256), 0, (hipStream_t)stream, src, Hp, Wp,
                       pad_top, pad_left, (unsigned char*)dst, H, W, bgr);
    return atmvfi::check_launch("frame_f32_to_u8");
}

extern "C" int atmvfi_final_residual(const float* it, const float* r, int r_ld, float* it_sum, float* it_clamped, int B,
                                      int H, int W, void* stream) {
    ATMVFI_REQUIRE(it && r && it_sum && it_clamped && B > 0 && H > 0 && W > 0 && r_ld >= 3, ATMVFI_EINVAL,
                   "final_residual: bad arguments");
    hipLaunchKernelGGL(final_residual_kernel, dim3(grid_for((long long)B * H * W)), dim3(256), 0, (hipStream_t)stream, it, r,
                       r_ld, it_sum, it_clamped, B, H, W);
    return atmvfi::check_launch("final_residual");
}

extern "C" int atmvfi_refine_tail(const float* contrib, int64_t contrib_plane, const float* bias, const float* slope, const float* it,
                                  float* it_sum, float* it_clamped, int B, int H, int W, void* stream) {
    ATMVFI_REQUIRE(contrib && it && it_sum && it_clamped && B > 0 && H > 0 && W > 0 && contrib_plane >= (int64_t)B * H * W, ATMVFI_EINVAL,
                   "refine_tail: bad arguments");
    hipLaunchKernelGGL(refine_tail_kernel, dim3(grid_for((long long)B * H * W)), dim3(256), 0, (hipStream_t)stream, contrib,
                       (long long)contrib_plane, bias, slope, it, it_sum, it_clamped, B, H, W);
    return atmvfi::check_launch("refine_tail");
}

extern "C" int atmvfi_ensemble_select(const float* loss0, const float* loss1, const float* loss2, const float* c0_l0, const float* c1_l0,
                                      const float* c0_l1, const float* c1_l1, const float* c0_l2, const float* c1_l2, float* out0, float* out1, int B,
                                      int64_t per_sample, void* stream) {
    ATMVFI_REQUIRE(loss0 && loss1 && loss2 && c0_l0 && c1_l0 && c0_l1 && c1_l1 && c0_l2 && c1_l2 && out0 && out1 && B > 0 && per_sample > 0,
                   ATMVFI_EINVAL, "ensemble_select: bad arguments");
    hipLaunchKernelGGL(ensemble_select_kernel, dim3(grid_for((long long)B * per_sample)), dim3(256), 0, (hipStream_t)stream, loss0, loss1, loss2,
                       c0_l0, c1_l0, c0_l1, c1_l1, c0_l2, c1_l2, out0, out1, B, (long long)per_sample);
    return atmvfi::check_launch("ensemble_select");
}

static int l1_mean_blocks(int64_t per_sample) {
    long long bx = (per_sample + 256 * 16 - 1) / (256 * 16);
    return (int)(bx > 1024 ? 1024 : bx);
}
extern "C" int64_t atmvfi_l1_mean_workspace_floats(int B, int64_t per_sample) {
    return (B > 0 && per_sample > 0) ? (int64_t)B * l1_mean_blocks(per_sample) : 0;
}
extern "C" int atmvfi_l1_mean(const float* a, const float* b, float* out, int B, int64_t per_sample, float* workspace, int64_t workspace_floats,
                              void* stream) {
    ATMVFI_REQUIRE(a && b && out && B > 0 && per_sample > 0, ATMVFI_EINVAL, "l1_mean: bad arguments");
    const int bx = l1_mean_blocks(per_sample);
    ATMVFI_REQUIRE(workspace && workspace_floats >= (int64_t)B * bx, ATMVFI_EINVAL,
                   "l1_mean: needs a workspace of atmvfi_l1_mean_workspace_floats(B, per_sample) = %lld floats", (long long)B * bx);
    hipLaunchKernelGGL(l1_partial_kernel, dim3((unsigned)bx, (unsigned)B), dim3(256), 0, (hipStream_t)stream, a, b, workspace,
                       (long long)per_sample);
    hipLaunchKernelGGL(l1_final_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, workspace, bx, out, (long long)per_sample);
    return atmvfi::check_launch("l1_mean");
}

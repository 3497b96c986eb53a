// Shared pieces of the 3x3 split-plane convolution kernels (conv3x3_planes.hip: two-accumulator kernel, launcher, split-K;
// conv3x3_planes_de.hip: the deferred-epilogue kernel).
#pragma once
#include "conv3_common.h"

namespace atmvfi {

struct Conv3PDev {
    const _Float16* in_hi;      // input planes, already advanced to the first 32-channel chunk of the view
    const _Float16* in_lo;
    long long in_rows;          // plane rows (> N*H*W; row N*H*W of every chunk is zero): chunk stride = in_rows * 32 halves
    int N, H, W, Cin;
    const _Float16* w_hi;
    const _Float16* w_lo;
    int wrows, cf, tail;
    int Cout;
    float* out;                 // optional fp32 NHWC view
    int out_ld;
    const float* bias;
    const float* prelu;
    _Float16* out_hi;           // optional plane sink (chunk major), channels out_c0 .. out_c0 + Cout
    _Float16* out_lo;
    long long plane_rows;
    int out_c0;
    const float* plane_prelu;   // optional PReLU applied to the plane copy only (the next layer's leading activation)
    _Float16* out_hi2;          // optional SECOND plane sink, raw (no PReLU of its own): a decoder map goes on both through the next
    _Float16* out_lo2;          // stage's leading PReLU (first sink) and as it is (the U-Net's strided convs read it)
    long long plane_rows2;
    int out_c02;
    int out_cmin;               // fp32 output: only channels >= out_cmin (multiple of 4) are stored
    int tiles_x, tiles_y, nblocks, tchunk;
    int vblocks;                // virtual blocks (tiles incl. XCD padding) walked by the persistent grid
    // the tile decode's divisors as multiply-shift pairs (launch_planes: q = (mul_hi(n, m) + n) >> s, exact for n < 2^31): dividing by a
    // kernel argument costs ~18 scalar instructions, and the decode of a workgroup's next-but-one tile -- five divisions -- sits on the
    // critical path of every tile boundary (1.3-1.5 k cycles per tile, tools/stamp_conv3p.py)
    unsigned dm_nblocks, ds_nblocks, dm_perimg, ds_perimg, dm_grp, ds_grp, dm_rows, ds_rows;
    // split-K (under-filled grids with long K, round 4): gridDim.y = ksplit workgroups per tile; split s takes the cps full chunks from
    // chunk s * cps on (the last one the rest and the tap-packed tail) and stores its raw fp32 sums at out + s * part_stride; a second
    // kernel adds the partial sums in split order and runs the epilogue.  1 = off.
    int ksplit, cps;
    long long part_stride;
    // fused read-out (refine_head.0 -> refine_head.1, network_base.py:257-260; WN = 2 or 4, one column block): the tile's activated
    // output -- still in registers, in the accumulator layout, which IS the B-operand layout of the next MFMA -- is multiplied by the
    // 27 x Cout matrix W2[(tap, o)][c] of the following 3-output 3x3 convolution; the 27 per-pixel "tap contributions" go to planar
    // fp32 h2_out[(tap * 3 + o) * h2_plane + pixel] and atmvfi_refine_tail adds each output pixel's nine shifted contributions.
    const _Float16* h2_w;       // [plane hi / lo][row tile 2][k-step WN/2][lane 64][8 halves], k order = this kernel's register order
    float* h2_out;
    long long h2_plane;
    int defer;                  // 1: multi-tile launches may take conv3x3_planes_de_kernel (wn with bit 4, include/atmvfi.h)
    unsigned long long* stamp;  // diagnostic builds only (ATMVFI_STAMP)
    int dbg;                    // diagnostic builds only: ATMVFI_P3_DBG bits switch pieces of the loop off (wrong results, timing only)
};

constexpr int HALO_PIX = HW_ * HW_;           // 324
constexpr int HALO_PLANE_PIECES = (HALO_PIX + 15) / 16;     // 21 one-KiB pieces (16 pixel rows x 64 B) per plane, 12 pad rows
constexpr int HALO_LO = HALO_PLANE_PIECES * 1024;            // byte offset of the lo plane inside a halo buffer
constexpr int HALO_BYTES = 2 * HALO_LO;

// weight ring depth: as many k-steps as fit beside the two halo buffers and the epilogue constants in 160 KiB, at most 5
// epilogue constants of a tile in LDS: bias, PReLU slope and the plane sink's own slope for its BN columns (three rows of BN floats)
constexpr int planes_const_floats(int BN) { return (3 * BN + 63) / 64 * 64; }
constexpr int ring_slots(int wn) {
    const int free_bytes = 160 * 1024 - 2 * HALO_BYTES - 2 * planes_const_floats(16 * wn) * 4;     // two halo buffers, two buffers of epilogue constants
    const int n = free_bytes / (2 * 16 * wn * 64);
    return n > 5 ? 5 : n;            // the static vmcnt counts of the k-loop assume a lookahead of at most 4 k-steps
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void dma16(const unsigned char* src, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// One LDS-DMA instruction per wave (4 bytes per lane): cst[c] = bias[n0 + c], cst[BN + c] = slope[n0 + c], cst[2 BN + c] = plane_slope[n0 + c]
// for the tile's BN columns, 0 / 1 / 1 where an array is absent or the column is past Cout.  (The plane sink's slopes used to be global loads
// inside the store loop: every one of them made hipcc wait for vmcnt(0), i.e. for the previous n-tile pair's STORES to be acknowledged --
// the sink epilogue of a 112-column tile took 9.4 k cycles against 5.1 k for fp32 rows, tools/stamp_conv3p.py.)
template <int BN>
__device__ __forceinline__ void dma_planes_consts(const float* bias, const float* slope, const float* plane_slope, int Cout, int n0, float* cst,
                                                  int wave, int lane) {
    constexpr int NPIECE = (3 * BN + 63) / 64;
    static_assert(NPIECE <= 8, "one piece per wave");
    const int piece = wave % NPIECE;
    const int t = piece * 64 + lane;
    const int row = t / BN;
    const int col = n0 + t - row * BN;
    const float* src = row == 0 ? bias : row == 1 ? slope : plane_slope;
    const float* p = (src && row < 3 && col < Cout) ? src + col : &kEpilogueDefaults[row == 0 ? 0 : 1];
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                     (__attribute__((address_space(3))) void*)(cst + piece * 64), 4, 0, 0);
}


// conv3x3_planes_de.hip: the deferred-epilogue kernel (bias, PReLU, one raw plane sink), tile width wn = 1..7
int launch_planes_de(int wn, const Conv3PDev& ds, int grid, size_t lds, hipStream_t s);

}  // namespace atmvfi

// Shared pieces of the split-precision 3x3 convolution (conv3x3_f16x3.hip: entry + packers, conv3x3_f16x3_row.hip: kernel).
#pragma once
#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

namespace atmvfi {
struct Conv3Dev {
    const float* in;
    int in_ld, N, H, W, Cin;
    const _Float16* w_hi;
    const _Float16* w_lo;
    int wrows;      // packed weight rows (multiple of 16)
    int cf;         // channels covered by full 32-channel chunks
    int tail;       // 1..8 channels in the tap-packed tail steps, 0 = none
    int cs, ktot;   // unused (row-major layout of an earlier revision)
    int Cout;
    float* out;
    int out_ld;
    const float* bias;
    const float* prelu;
    // optional second output: the result again as split planes (chunk-major, common.h RowSink), through its own per-channel
    // PReLU -- the form the next layer's LDS-DMA GEMM reads (decoder: conv -> [PReLU] -> ConvTranspose2d)
    _Float16* out_hi;
    _Float16* out_lo;
    long long plane_rows;
    const float* plane_prelu;
    int tiles_x, tiles_y;
    int nblocks;    // column blocks per spatial tile (set by the launcher)
    int legacy_order;   // A/B switch (ATMVFI_LEGACY_ORDER=1): round-robin tiles over XCDs
    int tchunk;     // spatial tiles per XCD = ceil(tiles / 8) (set by the launcher)
    int force_schedule, force_wn;   // per-call overrides: schedule -1 = cost model, 0 = row, 1 = half; wn 0 = cost model, 1..8
    unsigned long long* stamp;   // diagnostic builds only (ATMVFI_STAMP)
};
// conv3x3_f16x3_row.hip: three taps (one kernel row) per stage, single-buffered halo
int launch_conv3x3_row(const Conv3Dev& d, int ntiles, hipStream_t stream);
}  // namespace atmvfi
using atmvfi::Conv3Dev;

constexpr int TW = 16, HW_ = TW + 2;     // output tile width and halo width; the tile height is 2 rows per wavefront (conv3x3_f16x3_row.hip)

__device__ __forceinline__ int swz64(int row) { return ((row >> 2) & 1) << 1; }

// x = hi + lo'/1024 with hi = fp16(x) and lo' = fp16((x - hi) * 1024): scaling keeps lo' a NORMAL fp16
// whenever hi is (|lo'| <= |x|), so the pair carries ~22 significand bits at any magnitude down to
// fp16's normal range.  Both conversions clamp to +-65504 first, so finite fp32 never becomes inf.
constexpr float LO_SCALE = 1024.0f;
constexpr float LO_UNSCALE = 1.0f / 1024.0f;
__device__ __forceinline__ _Float16 sat_half(float v) { return (_Float16)fminf(fmaxf(v, -65504.0f), 65504.0f); }

__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, f16x8& hi, f16x8& lo) {
    const f32x2 x[4] = {{a.x, a.y}, {a.z, a.w}, {b.x, b.y}, {b.z, b.w}};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f16x2 h, l;
        split_pair(x[e], h, l);
        hi[2 * e] = h.x;
        hi[2 * e + 1] = h.y;
        lo[2 * e] = l.x;
        lo[2 * e + 1] = l.y;
    }
}

// Issue order of one stage's fragment reads and the wait each group needs.  Groups n = 0 .. NG-1 = (k-step n / WN, n-tile n % WN).
// Stage start: X(0) [4 reads], W(0) [2], W(1) [2].  Group n, before its MFMAs: W(n+2) [2], then (DX: two activation register sets)
// X(t+1) [4] if n % WN == JX; with one set (the 8-tile kernel has no 16 registers to spare) X(t+1) goes out right AFTER the MFMAs
// of the k-step's last group.  LDS returns in order, so wait(n) = number of reads issued after the youngest one group n consumes.
template <int WN, int TAPS, bool DX>
struct FragPipe {
    static constexpr int NG = TAPS * WN;
    static constexpr int JX = !DX ? WN - 1 : (WN >= 3 ? WN - 3 : 0);
    static constexpr int w_off(int n, int BN) { return ((n / WN) * BN + (n % WN) * 16) * 32 * 2; }      // bytes from the stage's hi plane
    static constexpr int wait(int n) {
        int issued = 4, seq_x[TAPS + 1] = {}, seq_w[NG + 3] = {};
        seq_x[0] = issued;
        for (int k = 0; k < 2 && k < NG; ++k) { issued += 2; seq_w[k] = issued; }
        for (int m = 0; m <= n; ++m) {
            if (m + 2 < NG) { issued += 2; seq_w[m + 2] = issued; }
            const bool x_here = m % WN == JX && m / WN + 1 < TAPS;
            if (x_here && DX) { issued += 4; seq_x[m / WN + 1] = issued; }
            if (m == n) break;
            if (x_here && !DX) { issued += 4; seq_x[m / WN + 1] = issued; }
        }
        const int need = seq_w[n] > seq_x[n / WN] ? seq_w[n] : seq_x[n / WN];
        return issued - need;
    }
};

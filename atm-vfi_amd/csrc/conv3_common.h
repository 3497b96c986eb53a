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
    int tiles_x, tiles_y;
    int nblocks;    // column blocks per spatial tile (set by the launcher)
    int legacy_order;   // A/B switch (ATMVFI_LEGACY_ORDER=1): round-robin tiles over XCDs
    int tchunk;     // spatial tiles per XCD = ceil(tiles / 8) (set by the launcher)
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
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 h = sat_half(x[e]);
        hi[e] = h;
        lo[e] = sat_half((x[e] - (float)h) * LO_SCALE);
    }
}


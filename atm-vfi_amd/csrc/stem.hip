// The encoder's full-resolution stem as ONE kernel (gfx950): feat_extracts.0.0 (3 -> C0, 3x3 + PReLU), feat_extracts.0.1 (C0 -> C0, 3x3 +
// PReLU) and feat_extracts.1.0 (C0 -> C1, 3x3 stride 2 + PReLU) of shared_feat_extraction (network_base.py:99-110, 342-352; conv() is
// :20-25), C0 / C1 = 24 / 48 (base) or 16 / 32 (lite).
//
// As three launches these layers move the two full-resolution C0-channel maps through HBM twice each (1.9 GB at 1080p for 71 GFLOP:
// 0.82 ms, a quarter of it compulsory); here a workgroup keeps them in LDS: per 8 x 16 tile of the half-resolution output it
//   A  loads the 21 x 37 pixel patch of the NHWC4 frames (zero outside the image) into LDS -- into the region that holds layer 3's
//      weights from phase C on: those are re-staged per tile by LDS-DMA under layer 2 (43 KB from L2; the LDS has no room for both);
//   B  computes the 19 x 35 patch of layer 1 on the matrix cores too: K = 27 is ONE k-step; the B fragment of a lane -- 8 of the 27
//      (tap, colour) values of its pixel -- is gathered from the LDS patch (eight ds_read_b32; gathered from global memory the eight
//      per-lane loads cost 140-230 cycles of issue EACH: 7-11 k cycles per tile in the address path, tools/stamp_stem.py) and split
//      in registers, the layer's A fragments stay in registers.  PReLU, ZERO outside the
//      image (that is layer 2's padding), split into fp16 hi / lo' and stored pixel-major (64-byte rows: C0 channels in 16-byte slots,
//      slot XOR-swizzled by row).  (First version: fp32 FMAs on the vector ALU, one pixel per lane, weights as scalar operands -- 648
//      weights per pixel block through ~80 SGPRs: 11 k cycles per 64 pixels of scalar-load latency, 22 k of a 41 k-cycle tile.)
//   C  layer 2 on the matrix cores with the f16x3 arithmetic of every other contraction (x = hi + lo'/1024, three
//      v_mfma_f32_16x16x32_f16 per product, two fp32 accumulators): K = 9 C0 ordered (tap, channel) and cut into k-steps of 32, so a lane
//      group's 8 k-values are 8 channels of ONE tap = one ds_read_b128 at (pixel + tap offset); an M-tile is 16 consecutive entries of
//      the 17 x 35 output patch linearised with the INPUT patch's row pitch (two junk columns per row), so that "pixel + tap offset"
//      is a plain row offset; each wave owns up to five M-tiles and walks the k-steps once, so a k-step's weight fragments (LDS,
//      staged once per persistent workgroup -- in registers they spill: 112 beside 80 accumulators) are read once per five tiles;
//   D  the results (bias, PReLU, zero outside the image) overwrite the layer-1 patch in LDS -- after a barrier; they wait in registers
//      until every wave has finished reading -- with the columns DE-INTERLEAVED by parity, so that the stride-2 taps of layer 3 are
//      again 16 consecutive rows per fragment;
//   E  layer 3 (stride 2) on the matrix cores, one output row of 16 pixels per wave, weights from LDS (staged once per workgroup), bias,
//      PReLU, split, and the result goes out as the split planes the next layer reads (chunk-major, include/atmvfi.h).
// Halo recompute: layer 1 x1.27, layer 2 x1.13 (incl. the junk columns).  HBM traffic: the frames in, the C1 planes out.
#include "common.h"

#ifdef ATMVFI_STAMP
static unsigned long long* g_stem_stamp = nullptr;
extern "C" void atmvfi_debug_set_stem_stamp_buffer(void* p) { g_stem_stamp = (unsigned long long*)p; }
#define ST_STAMP(i) do { if (a.stamp) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tph[i] += t_ - tlast; tlast = t_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define ST_STAMP(i) do { } while (0)
#endif

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct StemArgs {
    const float* x;                       // [F, H, W, 4] fp32 (channel 3 unused)
    int F, H, W, Ho, Wo;
    const _Float16* w1h;                  // [16 * J2][32]: k = (ky * 3 + kx) * 3 + ci, zero for k >= 27
    const _Float16* w1l;
    const float* b1;                      // padded like b2 / p2
    const float* p1;
    const _Float16* w2h;                  // [NS][16 * J2][32] k-step major, k = (ky * 3 + kx) * C0 + c
    const _Float16* w2l;
    const float* b2;                      // padded to 16 * J2 (bias 0 / slope 1 in the padding)
    const float* p2;
    const _Float16* w3h;                  // [NS][C1][32]
    const _Float16* w3l;
    const float* b3;
    const float* p3;
    _Float16* out_hi;
    _Float16* out_lo;
    long long plane_rows;
    int tiles_x, tiles_y, ntiles;
    unsigned long long* stamp;            // diagnostic build only: per (workgroup, wave) phase cycles
};

constexpr int TH = 8, TW = 16;                             // output tile (half resolution)
constexpr int E0H = 2 * TH + 1, E0W = 2 * TW + 1;          // layer-2 output patch 17 x 33
constexpr int AH = E0H + 2, AW = E0W + 2;                  // layer-1 output patch 19 x 35
constexpr int IH = AH + 2, IW = AW + 2;                    // frame patch 21 x 37
constexpr int NPA = AH * AW;                               // 665
constexpr int NM2 = (16 * AW + E0W + 15) / 16;             // 38 M-tiles of layer 2 (entries p = y * AW + x, y < 17)
constexpr int ACT_ROWS = 16 * NM2 + 2 * AW + 2 + 6;        // 686: every row a (junk) fragment can touch
constexpr int ACT_BYTES = ACT_ROWS * 64;
constexpr int HALF = (E0W + 1) / 2;                        // 17 even columns, then 16 odd ones
constexpr float LO_UNSCALE = 1.0f / 1024.0f;

// 16-byte slot swizzle of the 64-byte LDS rows: conflict-free for a weight fragment's ds_read_b128 (16 consecutive rows, slot = lane
// group), 1.36 cycles per conflict-free cycle for the activation fragments (the lane groups of one read sit at different taps) and
// 2-way for the epilogues' 8-byte stores of 16 consecutive rows (4-way with the contraction engines' ((row >> 2) & 1) << 1):
// brute force over the gfx950 lane groups, tools/lds_banks.py
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 3; }
__device__ __forceinline__ f32x4 prelu4(f32x4 v, const f32x4 s) {
    v.x = v.x > 0.f ? v.x : s.x * v.x;
    v.y = v.y > 0.f ? v.y : s.y * v.y;
    v.z = v.z > 0.f ? v.z : s.z * v.z;
    v.w = v.w > 0.f ? v.w : s.w * v.w;
    return v;
}

template <int C0, int C1>
__global__ __launch_bounds__(512, 2) void stem_kernel(const StemArgs a) {
    static_assert(C0 % 8 == 0 && C1 % 16 == 0 && C0 <= 32, "channel counts");
    constexpr int NS = (9 * C0 + 31) / 32, J2 = (C0 + 15) / 16, J3 = C1 / 16, SPP = C0 / 8;     // k-steps, n-tiles, 16-byte slots per pixel
    constexpr int W3_BYTES = NS * C1 * 64, W2_BYTES = NS * 16 * J2 * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const act_hi = smem;
    unsigned char* const act_lo = smem + ACT_BYTES;
    unsigned char* const w3h_l = smem + 2 * ACT_BYTES;
    unsigned char* const w3l_l = w3h_l + W3_BYTES;
    unsigned char* const w2h_l = w3l_l + W3_BYTES;
    unsigned char* const w2l_l = w2h_l + W2_BYTES;
    float* const cst = reinterpret_cast<float*>(w2l_l + W2_BYTES);      // b1 p1 b2 p2 (16 J2 floats each), b3 p3 (C1 each)
    f32x4* const img = reinterpret_cast<f32x4*>(w3h_l);                 // phases A-B: the frame patch, in layer 3's weight region
    static_assert(IH * IW * 16 <= 2 * W3_BYTES, "the frame patch must fit the layer-3 weight region");
    constexpr int R2 = 16 * J2;
    fp16_saturate_on();

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;

    // ---- once per workgroup: layer 2's weights into LDS (row = k-step * couts + cout, 64 bytes = 32 k-values, 16-byte slot swizzled by
    // row: an A fragment -- lane (r, g): k = 8 g .. 8 g + 7 of cout 16 j + r -- is one conflict-free ds_read_b128).  Layer 3's come per
    // tile (phase C), in the same layout.
    for (int i = tid; i < NS * 16 * J2 * 4; i += 512) {
        const int row = i >> 2, sl = i & 3;
        const int dst = row * 64 + ((sl ^ swz(row)) << 4);
        *reinterpret_cast<f16x8*>(w2h_l + dst) = *reinterpret_cast<const f16x8*>(a.w2h + (long long)row * 32 + sl * 8);
        *reinterpret_cast<f16x8*>(w2l_l + dst) = *reinterpret_cast<const f16x8*>(a.w2l + (long long)row * 32 + sl * 8);
    }
    for (int i = tid; i < 4 * R2 + 2 * C1; i += 512) {
        const float* src = i < R2 ? a.b1 + i : i < 2 * R2 ? a.p1 + (i - R2) : i < 3 * R2 ? a.b2 + (i - 2 * R2) : i < 4 * R2 ? a.p2 + (i - 3 * R2)
                           : i < 4 * R2 + C1 ? a.b3 + (i - 4 * R2) : a.p3 + (i - 4 * R2 - C1);
        cst[i] = *src;
    }
    // layer 1: per lane the source of each of its 8 k-values: k = 8 g + e = (tap, colour)
    int k1[8];                         // byte offset inside the LDS patch relative to the pixel's own top-left tap, -1: k >= 27 (zero)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * g + e;
        const int tap = k < 27 ? k / 3 : 4, ci = k < 27 ? k - 3 * (k / 3) : 0;
        k1[e] = k < 27 ? ((tap / 3) * IW + tap % 3) * 16 + ci * 4 : -1;
    }
    // per lane and k-step: row offset of the lane group's tap and its 16-byte slot, for layer 2 (pitch AW) and for layer 3 (pitch
    // E0W, columns de-interleaved by parity).  k-values past 9 C0 meet zero weights: any valid address will do.
    // (packed into one register per k-step: row offset of layer 2 | row offset of layer 3 << 10 | slot << 20)
    int kst[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int i = 4 * s + g;
        int tap = i / SPP;
        int sl = i - tap * SPP;
        if (tap > 8) { tap = 8; sl = 0; }
        const int ky = tap / 3, kx = tap - 3 * ky;
        kst[s] = (ky * AW + kx) | ((ky * E0W + (kx & 1) * HALF + (kx >> 1)) << 10) | (sl << 20);
    }

    // the frame patch of a tile: two 16-byte pixel records per thread, zero outside the image
    f32x4 nxt[2];
    auto request_patch = [&](int tt) {
        int q = tt;
        const int tx = q % a.tiles_x; q /= a.tiles_x;
        const int ty = q % a.tiles_y;
        const int n = q / a.tiles_y;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 512 * i;
            const int iy = idx / IW, ix = idx - iy * IW;
            const int gy = 2 * ty * TH - 3 + iy, gx = 2 * tx * TW - 3 + ix;
            const bool ok = idx < IH * IW && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            nxt[i] = *reinterpret_cast<const f32x4*>(a.x + (ok ? (unsigned)(((n * a.H + gy) * a.W + gx) * 4) : 0u));
            if (!ok) nxt[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    if ((int)blockIdx.x < a.ntiles) request_patch(blockIdx.x);
#ifdef ATMVFI_STAMP
    unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime(), ntl = 0;
#endif
    for (int t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
        int q = t;
        const int tx = q % a.tiles_x; q /= a.tiles_x;
        const int ty = q % a.tiles_y;
        const int n = q / a.tiles_y;
        const int oy0 = ty * TH, ox0 = tx * TW;            // output tile origin (half resolution)
        const int fy0 = 2 * oy0, fx0 = 2 * ox0;            // the same point at full resolution

        __syncthreads();            // the previous tile's layer-3 reads (patch buffer, weights) are done; first tile: layer 2's weights are staged
        ST_STAMP(0);

        // layer 1's A fragments (4 KB, L2-resident): per tile -- as kernel-long registers they push the kernel into spills whose
        // reloads (vector-memory operations, in-order vmcnt) then wait for the next tile's patch loads
        f16x8 w1h[J2], w1l[J2];
        {
            int rl0 = r, gl0 = g;
            asm volatile("" : "+v"(rl0), "+v"(gl0));
#pragma unroll
            for (int j = 0; j < J2; ++j) {
                w1h[j] = *reinterpret_cast<const f16x8*>(a.w1h + (16 * j + rl0) * 32 + 8 * gl0);
                w1l[j] = *reinterpret_cast<const f16x8*>(a.w1l + (16 * j + rl0) * 32 + 8 * gl0);
            }
        }
        // ---- A: frame patch, origin (fy0 - 3, fx0 - 3): requested one tile ahead (under layer 3 of the previous tile), stored here
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (tid + 512 * i < IH * IW) img[tid + 512 * i] = nxt[i];
        __syncthreads();

        // ---- B: layer 1 on the patch with origin (fy0 - 2, fx0 - 2): M-tiles of 16 patch pixels dealt to the waves
        constexpr int NM1 = (NPA + 15) / 16, MPW = (NM1 + 7) / 8;          // 42 M-tiles, up to 6 per wave
        float v[MPW][8];
        // (per-lane values are laundered through empty asm statements here: hipcc otherwise decodes the M-tiles' patch coordinates ONCE
        // per kernel, registers that it then spills and reloads in every tile)
        int rl = r;
        asm volatile("" : "+v"(rl));
#pragma unroll
        for (int i = 0; i < MPW; ++i) {
            const int mt = wave + 8 * i;
            const int p = mt * 16 + rl;
            const int ci_ = p < NPA ? p : NPA - 1;
            const int y = ci_ / AW, x = ci_ - y * AW;
            const unsigned char* pb = reinterpret_cast<const unsigned char*>(img) + (y * IW + x) * 16;     // the pixel's tap (0, 0)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = *reinterpret_cast<const float*>(pb + (k1[e] < 0 ? 0 : k1[e]));
                v[i][e] = k1[e] < 0 ? 0.f : t;
            }
        }
#ifdef ATMVFI_STAMP
        ST_STAMP(6);                                                  // gathers issued
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ST_STAMP(7);                                                  // ... and landed
#endif
#pragma unroll
        for (int i = 0; i < MPW; ++i) {
            const int mt = wave + 8 * i;
            if (mt < NM1) {                                          // wave-uniform
                const int p = mt * 16 + rl;
                const int ci_ = p < NPA ? p : NPA - 1;
                const int y = ci_ / AW, x = ci_ - y * AW;
                const int gy = fy0 - 2 + y, gx = fx0 - 2 + x;
                f16x2 h[4], l[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) split_pair((f32x2){v[i][2 * e], v[i][2 * e + 1]}, h[e], l[e]);
                const f16x8 xh = (f16x8){h[0].x, h[0].y, h[1].x, h[1].y, h[2].x, h[2].y, h[3].x, h[3].y};
                const f16x8 xl = (f16x8){l[0].x, l[0].y, l[1].x, l[1].y, l[2].x, l[2].y, l[3].x, l[3].y};
                const bool inside = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                const int sw = swz(p);
#pragma unroll
                for (int j = 0; j < J2; ++j) {
                    const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
                    f32x4 cor = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1l[j], xh, z, 0, 0, 0);
                    const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1h[j], xh, z, 0, 0, 0);
                    cor = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1h[j], xl, cor, 0, 0, 0);
                    const int c = 16 * j + 4 * g;
                    f32x4 o = acc + cor * LO_UNSCALE + *reinterpret_cast<const f32x4*>(cst + c);
                    o = prelu4(o, *reinterpret_cast<const f32x4*>(cst + R2 + c));
                    if (!inside) o = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (c < C0 && p < NPA) {
                        f16x2 h0, l0, h1, l1;
                        split_pair((f32x2){o.x, o.y}, h0, l0);
                        split_pair((f32x2){o.z, o.w}, h1, l1);
                        const int dst = p * 64 + (((c >> 3) ^ sw) << 4) + ((c & 4) << 1);
                        *reinterpret_cast<f16x4*>(act_hi + dst) = (f16x4){h0.x, h0.y, h1.x, h1.y};
                        *reinterpret_cast<f16x4*>(act_lo + dst) = (f16x4){l0.x, l0.y, l1.x, l1.y};
                    }
                }
            }
        }
        ST_STAMP(1);
        __syncthreads();
        ST_STAMP(2);
        // layer 3's weights over the frame patch (every wave has gathered): 1 KiB LDS-DMA pieces = 16 rows x 64 B, the slot swizzle on
        // the SOURCE address (the LDS side of a piece is lane-linear); waited for before phase E
        for (int pc = wave; pc < 2 * (W3_BYTES / 1024); pc += 8) {
            const bool lo = pc >= W3_BYTES / 1024;
            const int piece = lo ? pc - W3_BYTES / 1024 : pc;
            const int row = piece * 16 + (lane >> 2);
            const unsigned char* src = reinterpret_cast<const unsigned char*>(lo ? a.w3l : a.w3h) + row * 64 + (((lane & 3) ^ swz(row)) << 4);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)((lo ? w3l_l : w3h_l) + piece * 1024), 16, 0, 0);
        }

        // ---- C: layer 2.  M-tile mt: entries p = 16 mt + r of the 17 x AW linearised output patch (origin (fy0 - 1, fx0 - 1))
        f32x4 ev[5][J2];
        {
            f32x4 acc[5][J2], cor[5][J2];
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < J2; ++j) {
                    acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    cor[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                f16x8 w2h[J2], w2l[J2];
#pragma unroll
                for (int j = 0; j < J2; ++j) {
                    const int wrow = s * 16 * J2 + 16 * j + r;
                    const int woff = wrow * 64 + ((g ^ swz(wrow)) << 4);
                    w2h[j] = *reinterpret_cast<const f16x8*>(w2h_l + woff);
                    w2l[j] = *reinterpret_cast<const f16x8*>(w2l_l + woff);
                }
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const int mt = wave + 8 * i;
                    if (mt < NM2) {                                  // wave-uniform
                        const int row = 16 * mt + r + (kst[s] & 1023);
                        const int off = row * 64 + (((kst[s] >> 20) ^ swz(row)) << 4);
                        const f16x8 xh = *reinterpret_cast<const f16x8*>(act_hi + off);
                        const f16x8 xl = *reinterpret_cast<const f16x8*>(act_lo + off);
#pragma unroll
                        for (int j = 0; j < J2; ++j) {
                            cor[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2l[j], xh, cor[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2h[j], xh, acc[i][j], 0, 0, 0);
                            cor[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2h[j], xl, cor[i][j], 0, 0, 0);
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int p = 16 * (wave + 8 * i) + r;
                const int y = p / AW, x = p - y * AW;
                const int gy = fy0 - 1 + y, gx = fx0 - 1 + x;
                const bool inside = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
#pragma unroll
                for (int j = 0; j < J2; ++j) {
                    const int c = 16 * j + 4 * g;
                    f32x4 v = acc[i][j] + cor[i][j] * LO_UNSCALE + *reinterpret_cast<const f32x4*>(cst + 2 * R2 + c);
                    v = prelu4(v, *reinterpret_cast<const f32x4*>(cst + 3 * R2 + c));
                    ev[i][j] = inside ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        ST_STAMP(3);
        __syncthreads();                                             // every wave has read its layer-1 rows

        // ---- D: layer-2 results over the same buffer, row = y * E0W + (x & 1) * HALF + (x >> 1)
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int mt = wave + 8 * i;
            const int p = 16 * mt + r;
            const int y = p / AW, x = p - y * AW;
            if (mt < NM2 && x < E0W && y < E0H) {
                const int row = y * E0W + (x & 1) * HALF + (x >> 1);
                const int sw = swz(row);
#pragma unroll
                for (int j = 0; j < J2; ++j) {
                    const int c = 16 * j + 4 * g;
                    if (c < C0) {
                        f16x2 h0, l0, h1, l1;
                        split_pair((f32x2){ev[i][j].x, ev[i][j].y}, h0, l0);
                        split_pair((f32x2){ev[i][j].z, ev[i][j].w}, h1, l1);
                        const int dst = row * 64 + (((c >> 3) ^ sw) << 4) + ((c & 4) << 1);
                        *reinterpret_cast<f16x4*>(act_hi + dst) = (f16x4){h0.x, h0.y, h1.x, h1.y};
                        *reinterpret_cast<f16x4*>(act_lo + dst) = (f16x4){l0.x, l0.y, l1.x, l1.y};
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's pieces of layer 3's weights have landed
        __syncthreads();
        ST_STAMP(4);

        // ---- E: layer 3, output row oy = wave, pixels ox = r: tap (ky, kx) reads patch pixel (2 oy + ky, 2 r + kx)
        if (t + (int)gridDim.x < a.ntiles) request_patch(t + gridDim.x);     // the next tile's frames, in flight under this phase
        {
            f32x4 acc[J3], cor[J3];
#pragma unroll
            for (int j = 0; j < J3; ++j) {
                acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                cor[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            const int rbase = 2 * wave * E0W + r;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int row = rbase + ((kst[s] >> 10) & 1023);
                const int off = row * 64 + (((kst[s] >> 20) ^ swz(row)) << 4);
                const f16x8 xh = *reinterpret_cast<const f16x8*>(act_hi + off);
                const f16x8 xl = *reinterpret_cast<const f16x8*>(act_lo + off);
#pragma unroll
                for (int j = 0; j < J3; ++j) {
                    const int wrow = s * C1 + 16 * j + r;
                    const int woff = wrow * 64 + ((g ^ swz(wrow)) << 4);
                    const f16x8 wh = *reinterpret_cast<const f16x8*>(w3h_l + woff);
                    const f16x8 wl = *reinterpret_cast<const f16x8*>(w3l_l + woff);
                    cor[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh, cor[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, acc[j], 0, 0, 0);
                    cor[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, cor[j], 0, 0, 0);
                }
            }
            const int gy = oy0 + wave, gx = ox0 + r;
            if (gy < a.Ho && gx < a.Wo) {
                const long long prow = ((long long)n * a.Ho + gy) * a.Wo + gx;
#pragma unroll
                for (int j = 0; j < J3; ++j) {
                    const int c = 16 * j + 4 * g;
                    f32x4 v = acc[j] + cor[j] * LO_UNSCALE + *reinterpret_cast<const f32x4*>(cst + 4 * R2 + c);
                    v = prelu4(v, *reinterpret_cast<const f32x4*>(cst + 4 * R2 + C1 + c));
                    f16x2 h0, l0, h1, l1;
                    split_pair((f32x2){v.x, v.y}, h0, l0);
                    split_pair((f32x2){v.z, v.w}, h1, l1);
                    const long long off = ((long long)(c >> 5) * a.plane_rows + prow) * 32 + (c & 31);
                    *reinterpret_cast<f16x4*>(a.out_hi + off) = (f16x4){h0.x, h0.y, h1.x, h1.y};
                    *reinterpret_cast<f16x4*>(a.out_lo + off) = (f16x4){l0.x, l0.y, l1.x, l1.y};
                }
            }
        }
        ST_STAMP(5);
#ifdef ATMVFI_STAMP
        ++ntl;
#endif
    }
#ifdef ATMVFI_STAMP
    if (a.stamp && lane == 0) {
        unsigned long long* o = a.stamp + ((long long)blockIdx.x * 8 + wave) * 8;
        for (int i = 0; i < 6; ++i) o[i] = tph[i];
        o[6] = ntl;
        o[7] = (tph[6] << 32) | (tph[7] & 0xffffffffull);
    }
#endif
}

template <int C0, int C1>
int launch_stem(const StemArgs& a, hipStream_t s) {
    constexpr int NS = (9 * C0 + 31) / 32;
    constexpr int J2 = (C0 + 15) / 16;
    constexpr size_t lds = 2 * (size_t)ACT_BYTES + 2 * (size_t)NS * C1 * 64 + 2 * (size_t)NS * 16 * J2 * 64 + (4 * 16 * J2 + 2 * C1) * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS budget");
    static_assert((NS * C1 * 64) % 1024 == 0, "layer-3 weight planes are whole 1 KiB DMA pieces");
    const hipError_t e = atmvfi::allow_dynamic_lds<stem_kernel<C0, C1>>(lds);
    ATMVFI_REQUIRE(e == hipSuccess, ATMVFI_ELAUNCH, "stem_fused: hipFuncSetAttribute: %s", hipGetErrorString(e));
    const int grid = std::min(a.ntiles, atmvfi::cu_count());
    hipLaunchKernelGGL((stem_kernel<C0, C1>), dim3((unsigned)grid), dim3(512), lds, s, a);
    return atmvfi::check_launch("stem_fused");
}

}  // namespace

extern "C" int atmvfi_stem_fused(const float* x, int F, int H, int W, int C0, int C1, const void* w1_hi, const void* w1_lo, const float* b1,
                                 const float* p1, const void* w2_hi, const void* w2_lo, const float* b2, const float* p2, const void* w3_hi,
                                 const void* w3_lo, const float* b3, const float* p3, void* out_hi, void* out_lo, int64_t out_plane_rows,
                                 void* stream) {
    ATMVFI_REQUIRE(x && w1_hi && w1_lo && b1 && p1 && w2_hi && w2_lo && b2 && p2 && w3_hi && w3_lo && b3 && p3 && out_hi && out_lo, ATMVFI_EINVAL,
                   "stem_fused: null pointer");
    ATMVFI_REQUIRE(F > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, ATMVFI_EINVAL, "stem_fused: bad shape F %d H %d W %d (even sizes)", F, H, W);
    ATMVFI_REQUIRE((C0 == 24 && C1 == 48) || (C0 == 16 && C1 == 32), ATMVFI_EINVAL,
                   "stem_fused: channel counts %d -> %d (the variants' 24 -> 48 and 16 -> 32 are built)", C0, C1);
    const long long rows = (long long)F * (H / 2) * (W / 2);
    ATMVFI_REQUIRE(out_plane_rows >= rows, ATMVFI_EINVAL, "stem_fused: %lld plane rows < %lld output pixels", (long long)out_plane_rows, rows);
    const void* al[] = {x, w1_hi, w1_lo, b1, p1, w2_hi, w2_lo, b2, p2, w3_hi, w3_lo, b3, p3, out_hi, out_lo};
    for (const void* p : al) ATMVFI_REQUIRE(atmvfi::aligned16(p), ATMVFI_EALIGN, "stem_fused: pointer not 16-byte aligned");
    StemArgs a;
    a.x = x; a.F = F; a.H = H; a.W = W; a.Ho = H / 2; a.Wo = W / 2;
    a.w1h = (const _Float16*)w1_hi; a.w1l = (const _Float16*)w1_lo; a.b1 = b1; a.p1 = p1;
    a.w2h = (const _Float16*)w2_hi; a.w2l = (const _Float16*)w2_lo; a.b2 = b2; a.p2 = p2;
    a.w3h = (const _Float16*)w3_hi; a.w3l = (const _Float16*)w3_lo; a.b3 = b3; a.p3 = p3;
    a.out_hi = (_Float16*)out_hi; a.out_lo = (_Float16*)out_lo; a.plane_rows = out_plane_rows;
    a.tiles_x = (a.Wo + TW - 1) / TW; a.tiles_y = (a.Ho + TH - 1) / TH;
    const long long nt = (long long)F * a.tiles_x * a.tiles_y;
    ATMVFI_REQUIRE(nt < (1ll << 31) && (long long)F * H * W * 4 < (1ll << 31), ATMVFI_EINVAL, "stem_fused: frames too large for 32-bit indices");
    a.ntiles = (int)nt;
    a.stamp = nullptr;
#ifdef ATMVFI_STAMP
    a.stamp = g_stem_stamp;
#endif
    return C0 == 24 ? launch_stem<24, 48>(a, (hipStream_t)stream) : launch_stem<16, 32>(a, (hipStream_t)stream);
}

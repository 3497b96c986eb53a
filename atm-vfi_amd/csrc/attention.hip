// Fused window attention for gfx950 (ATMFormer cross-attention + motion, Swin self-attention).
//
// One workgroup per (window, head); one wavefront per 16-query tile.
//   S^T = K Q^T   : MFMA A = K fragment (LDS, ds_read_b128), B = Q fragment (global, all chunks requested up front)
//                   -> lane (r = lane&15, g = lane>>4) holds S[q = 16w + r][key = 16kt + 4g + e]
//   softmax       : lane-local over its 4*NT keys, then two __shfl_xor (16, 32) across the 4 lane groups
//   motion        : sum_k P[q,k] * (k_xy - q_xy) from the same registers (no relative_coord table)
//   O^T = V^T P^T : MFMA A = V^T fragment (LDS holds V transposed: ds_read_b128), B = P^T = the S accumulators as they stand
//                   (key index on the register axis = the k axis of the next MFMA; no LDS round trip)
//                   -> lane holds O[q][d = 16dt + 4g + e]: one 16-byte store per d-tile.
// The N x N attention matrix and the 2 x N x N motion product of the reference
// (attention.py:192-208) are never materialised.
#include "common.h"

#include <math.h>

namespace {

// LDS images, both read as MFMA A-operand fragments by ONE ds_read_b128 per lane and 4 k-steps:
//   Ks [NPAD keys][PK slots of 16 B]   K[key][d],   fragment (key tile kt, d chunk c): row 16 kt + r, slot 4 c + g
//   Vt [DPAD d   ][PV slots of 16 B]   V^T[d][key], fragment (d tile dt, key tile kt): row 16 dt + r, slot 4 kt + g
// with the slot XOR-swizzled by (row & 15): the 16 lanes ds_read_b128 serves per cycle ({0-3,12-15,20-27}, ...) are rows
// {0-3,12-15} at slot s and rows {4-11} at slot s + 1, and (s ^ {0-3,12-15}) u ((s+1) ^ {4-11}) are 16 different slots = all 64
// banks once.  (The first version kept V row-major and fed the PV MFMAs from scalar ds_read_b32 column reads, and its K rows had a
// pitch of hd + 4 floats: SQ_LDS_BANK_CONFLICT was 75 % of SQ_INSTS_LDS, profiles/r01_lds_counters.txt.)
__device__ __forceinline__ int slots16(int n) { return (n + 15) & ~15; }

template <int NT>
__global__ __launch_bounds__(64 * NT) void window_attn_kernel(
    const float* __restrict__ qkv, const RowSink out, float* __restrict__ motion,
    const int* __restrict__ labels, int N, int nW, int ws, int heads, int hd, int C, int Bw, int kv_shift,
    float scale) {
    fp16_saturate_on();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NPAD = NT * 16;
    const int nchunks = (hd + 15) >> 4;          // 16-wide d chunks (QK^T) = 16-row d tiles (PV)
    const int DPAD = 16 * nchunks;
    const int PK = slots16(DPAD / 4), PV = slots16(NPAD / 4);
    float* Ks = smem;                            // [NPAD][PK * 4]
    float* Vt = Ks + NPAD * PK * 4;              // [DPAD][PV * 4]
    int* Ls = reinterpret_cast<int*>(Vt + DPAD * PV * 4);

    const int tid = threadIdx.x;
    // XCD-aware order: blocks i and i + 8 share an XCD (and its L2), so the `heads` workgroups of one window run back to back on
    // ONE XCD: a head's slice of a qkv row is hd * 4 bytes (192 B at hd = 48), i.e. it shares cache lines with its neighbours,
    // and with the heads of a window dealt round-robin over the eight L2s every one of them fetched those lines from HBM again
    // (PMC: 1.34x the algorithmic read bytes).
    const int slot = blockIdx.x >> 3;
    const int b = (slot / heads) * 8 + (blockIdx.x & 7);
    const int h = slot % heads;
    if (b >= Bw) return;
    const int bk = (b + kv_shift) % Bw;
    const int C3 = 3 * C;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int r = lane & 15;
    const int g = lane >> 4;
    const int q = 16 * w + r;
    const bool qok = q < N;

    // ---- Q fragments of this wave, requested before anything else: they land while K and V are staged ----
    constexpr int MAXC = NT >= 13 ? 4 : 8;       // hd <= 128 (<= 64 for the 13..16-tile windows: 1024 threads leave 128 registers)
    f32x4 qf[MAXC];
    const float* qrow = qkv + ((long long)b * N + (qok ? q : 0)) * C3 + h * hd;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        qf[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int d = 16 * c + 4 * g;
        if (c < nchunks && qok && d < hd) qf[c] = *reinterpret_cast<const f32x4*>(qrow + d);
    }

    // ---- stage K (row-major) and V (transposed) of this head; zero for padded keys / head-dim padding.  All global loads of
    // the pass are requested before the first LDS write (unconditional loads from clamped addresses, zeroed by selects: a
    // predicated load makes hipcc wait for each one on the spot), so a workgroup pays ONE memory latency for its K/V tile. ----
    const int dg = DPAD >> 2;
    constexpr int UNR = 6;                                   // local: 64 keys x 12 groups / 256 threads = 3; global: 144 x 24 / 576 = 6
    for (int base = 0; base < NPAD * dg; base += 64 * NT * UNR) {
        f32x4 kv[UNR], vv[UNR];
        int keyv[UNR], d4v[UNR];
        bool okv[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int idx = base + u * 64 * NT + tid;
            const int key = idx / dg;
            const int d4 = idx - key * dg;
            keyv[u] = key;
            d4v[u] = d4;
            okv[u] = key < N && (d4 << 2) < hd;
            const float* p = qkv + ((long long)bk * N + (okv[u] ? key : 0)) * C3 + C + h * hd + (okv[u] ? (d4 << 2) : 0);
            kv[u] = *reinterpret_cast<const f32x4*>(p);
            vv[u] = *reinterpret_cast<const f32x4*>(p + C);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int key = keyv[u], d4 = d4v[u], d = d4 << 2;
            if (key < NPAD) {
                const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
                const f32x4 k4 = okv[u] ? kv[u] : z, v4 = okv[u] ? vv[u] : z;
                *reinterpret_cast<f32x4*>(Ks + (key * PK + (d4 ^ (key & 15))) * 4) = k4;
                const int ks = key >> 2, ke = key & 3;
#pragma unroll
                for (int e = 0; e < 4; ++e) Vt[((d + e) * PV + (ks ^ ((d + e) & 15))) * 4 + ke] = v4[e];
            }
        }
    }
    for (int idx = tid; idx < NPAD; idx += 64 * NT)
        Ls[idx] = (labels && idx < N) ? labels[(long long)(b % nW) * N + idx] : 0;
    __syncthreads();

    // ---- S^T = K Q^T ----
    f32x4 s[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        if (c < nchunks) {
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + ((16 * kt + r) * PK + ((4 * c + g) ^ r)) * 4);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[ks], qf[c][ks], s[kt], 0, 0, 0);
            }
        }
    }

    // ---- scale + mask + softmax over keys (attention.py:192-200) ----
    const int lab_q = Ls[qok ? q : 0];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        const int4 lk = *reinterpret_cast<const int4*>(Ls + 16 * kt + 4 * g);
        const int lks[4] = {lk.x, lk.y, lk.z, lk.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int key = 16 * kt + 4 * g + e;
            float v = s[kt][e] * scale;
            if (labels && lks[e] != lab_q) v += -100.0f;
            if (key >= N) v = -INFINITY;
            s[kt][e] = v;
            mx = fmaxf(mx, v);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float p = expf(s[kt][e] - mx);
            s[kt][e] = p;
            sum += p;
        }
    const float inv_ws = 1.0f / (float)ws;        // key / ws for key < 256, ws <= 16: floor((key + 0.5) / ws) is exact in fp32
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    const float qy = floorf(((float)q + 0.5f) * inv_ws), qx = (float)q - qy * (float)ws;
    float mox = 0.f, moy = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int key = 16 * kt + 4 * g + e;
            const float p = s[kt][e] * inv;
            s[kt][e] = p;
            const float ky = floorf(((float)key + 0.5f) * inv_ws);
            mox += p * (((float)key - ky * (float)ws) - qx);
            moy += p * (ky - qy);
        }
    if (motion) {
        mox += __shfl_xor(mox, 16);
        mox += __shfl_xor(mox, 32);
        moy += __shfl_xor(moy, 16);
        moy += __shfl_xor(moy, 32);
        if (g == 0 && qok) {
            float* mp = motion + (((long long)b * N + q) * heads + h) * 2;
            mp[0] = mox;
            mp[1] = moy;
        }
    }

    // ---- O^T = V^T P^T: the S accumulators as they stand are the B operand (key index on the register axis) ----
    const long long orow = (long long)b * N + (qok ? q : 0);
    for (int dt = 0; dt < nchunks; ++dt) {
        f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            const f32x4 vf = *reinterpret_cast<const f32x4*>(Vt + ((16 * dt + r) * PV + ((4 * kt + g) ^ r)) * 4);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                o = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[ks], s[kt][ks], o, 0, 0, 0);
        }
        const int d = 16 * dt + 4 * g;
        if (qok && d < hd) sink_store4(out, orow, h * hd + d, o);
    }
}

// the A operand of one 16x16x32 step out of a row-major [key][d] fp16 image: two transposing reads (keys +0..3 and +16..19 of the
// lane group's block, 16 d columns), each 4 halves per lane
typedef __fp16 h16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ f16x8_t tr_pair(unsigned a0, unsigned a1) {
    const h16x4_t x = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h16x4_t*)(unsigned long long)a0);
    const h16x4_t y = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h16x4_t*)(unsigned long long)a1);
    struct Two { h16x4_t x, y; };
    return __builtin_bit_cast(f16x8_t, (Two){x, y});
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same attention on the f16x3 arithmetic of the contraction engines (x = hi + lo'/1024 in fp16, three
// v_mfma_f32_16x16x32_f16 per product into two fp32 accumulators: 1/4 of the fp32-MFMA cycles, ~22 significand bits).
//   staging       : K and V of the head are converted while they are staged, ROW-MAJOR, as fp16 hi / lo' images (no transposing
//                   store: the old kernel's scalar ds_write_b32 transpose of V was 2.7 conflict cycles per LDS instruction);
//                   K rows are 16-byte slots XOR-swizzled for the ds_read_b128 lane groups, V rows have a pitch that is an odd
//                   multiple of 32 bytes for the transposed reads (both images: 0 conflicts by the bank rules, tools/lds_banks.py)
//   S^T = K Q^T   : A = K fragment (ds_read_b128: 8 consecutive d of one key), B = Q fragment (global, split in registers)
//                   -> lane (r = lane&15, g = lane>>4) holds S[q = 16w + r][key = 16kt + 4g + e], exactly as the fp32 kernel
//   softmax / motion: unchanged, fp32, lane-local + two __shfl_xor
//   O^T = V^T P^T : B = P^T straight from the S registers: k-step kt2 takes tiles 2 kt2 and 2 kt2 + 1, element j of lane group g is
//                   key 32 kt2 + 16 (j>>2) + 4 g + (j&3); A = V^T fragment in that same key order = two ds_read_b64_tr_b16 (the
//                   gfx950 transposing read: 4 keys x 16 d per 16 lanes, delivered column-major) of the row-major V image.
#ifdef ATMVFI_STAMP
__device__ unsigned long long* g_attn_stamp = nullptr;   // diagnostic build only: phase cycles of workgroup 0, wave 0 (tools/stamp_attn.py)
#define AT_STAMP(i)                                                     \
    do {                                                                \
        if (stamping) {                                                 \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
            tacc[i] += t_ - tlast;                                      \
            tlast = t_;                                                 \
        }                                                               \
    } while (0)
#else
#define AT_STAMP(i) do { } while (0)
#endif
// Persistent: a workgroup walks (window, head) items v = blockIdx.x, + gridDim.x, ... (grid a multiple of 8, so v & 7 stays the XCD and
// the heads of a window still run together on one XCD).  The K / V rows of the NEXT item are requested into registers before the
// current item's products start and converted into the LDS images after them (8 160 one-shot workgroups spent most of their life
// waiting for their first loads and for the dispatcher); everything that does not depend on the item -- the zero padding of both
// images, the key-position columns -- is written once.
template <int NT, int DCH>        // DCH = ceil(hd / 32): 32-wide d chunks of the QK^T product
__global__ __launch_bounds__(64 * NT, (NT <= 4 ? 3 : NT <= 8 ? 2 : NT <= 12 ? 3 : 4)) void window_attn_x3_kernel(
    const float* __restrict__ qkv, const RowSink out, float* __restrict__ motion,
    const int* __restrict__ labels, int N, int nW, int ws, int heads, int hd, int C, int Bw, int kv_shift,
    float scale, int vitems) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    fp16_saturate_on();
    constexpr int NPAD = NT * 16, KT2 = (NT + 1) / 2, VROWS = 32 * KT2, T = 64 * NT;
    // windows of more than 128 tokens (9+ waves) have no registers for the next item's rows: one item per workgroup there (the
    // loop below runs once and the compiler sees it)
    constexpr bool PERSIST = NT <= 8, PREFETCH = PERSIST, PREFETCH_Q = PERSIST;
    constexpr int UNR = 2 * DCH;                  // 4-float units of K and of V per thread: N hd / 4 <= 64 NT * 2 DCH
    // With a motion output the V image carries four more columns hd .. hd + 3 = (key x, key y, 0, 0): the expected key position
    // sum_k P[q,k] k_xy falls out of the PV product as two more output columns instead of ~9 VALU instructions per (q, key).
    constexpr int D32 = DCH;
    const int DT = (hd + (motion ? 4 : 0) + 15) >> 4;
    constexpr int SPK = D32 <= 1 ? 4 : D32 == 2 ? 8 : 16;     // 16-byte slots per K row and plane (power of two: XOR swizzle)
    constexpr int KS = SPK * 16;                              // K row pitch, bytes
    const int VS = 32 * DT + ((DT & 1) ? 0 : 32);         // V row pitch, bytes: odd multiple of 32
    unsigned char* Kh = reinterpret_cast<unsigned char*>(smem);
    unsigned char* Kl = Kh + NPAD * KS;
    unsigned char* Vh = Kl + NPAD * KS;
    unsigned char* Vl = Vh + VROWS * VS;
    int* Ls = reinterpret_cast<int*>(Vl + VROWS * VS);

    const int tid = threadIdx.x;
    const int C3 = 3 * C;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int r = lane & 15;
    const int g = lane >> 4;
    const int q = 16 * w + r;
    const bool qok = q < N;
    const float inv_ws = 1.0f / (float)ws;        // key / ws for key < 256, ws <= 16: floor((key + 0.5) / ws) is exact in fp32

    // ---- once: zeros where no staged value ever lands (padded keys, K's head-dim padding), then the key-position columns.  A
    // persistent workgroup clears both images whole; a one-item workgroup (145 KiB at ws 12 / hd 84) only those regions. ----
    {
        const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
        const f16x4 z = (f16x4){(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        if constexpr (PERSIST) {
            const int words16 = (2 * NPAD * KS + 2 * VROWS * VS) >> 4;
            for (int idx = tid; idx < words16; idx += T) reinterpret_cast<f32x4*>(smem)[idx] = z4;
            __syncthreads();
        } else {
            const int dg0 = hd >> 2, kpad = 8 * DCH - dg0;            // staged / padding 4-float units of a K row
            auto kzero = [&](int row, int d4) {
                const int sw = SPK == 16 ? (row & 15) : SPK == 8 ? ((row >> 1) & 7) : ((0x1320 >> (((row >> 2) & 3) * 4)) & 3);
                const int off = row * KS + ((((d4 >> 1) ^ sw) & (SPK - 1)) << 4) + ((d4 & 1) << 3);
                *reinterpret_cast<f16x4*>(Kh + off) = z;
                *reinterpret_cast<f16x4*>(Kl + off) = z;
            };
            for (int idx = tid; idx < N * kpad; idx += T) kzero(idx / kpad, dg0 + idx % kpad);
            for (int idx = tid; idx < (NPAD - N) * 8 * DCH; idx += T) kzero(N + idx / (8 * DCH), idx % (8 * DCH));
            const int vtail16 = ((VROWS - N) * VS) >> 4;              // V rows of the padded keys (VS % 32 == 0)
            for (int idx = tid; idx < vtail16; idx += T) {
                reinterpret_cast<f32x4*>(Vh + N * VS)[idx] = z4;
                reinterpret_cast<f32x4*>(Vl + N * VS)[idx] = z4;
            }
        }
        if (motion && tid < N) {                  // small integers: exact in fp16, lo' = 0
            const float ky = floorf(((float)tid + 0.5f) * inv_ws), kx = (float)tid - ky * (float)ws;
            *reinterpret_cast<f16x4*>(Vh + tid * VS + (hd << 1)) = (f16x4){(_Float16)kx, (_Float16)ky, (_Float16)0.f, (_Float16)0.f};
            *reinterpret_cast<f16x4*>(Vl + tid * VS + (hd << 1)) = z;
        }
    }

    // ---- staging units: 4 floats (row, d4) of the head's K and V, item independent per thread ----
    const int dgr = hd >> 2;                      // units per row (hd % 4 == 0)
    const int units = N * dgr;
    const float inv_dgr = 1.0f / (float)dgr;
    constexpr int ksw_mask = SPK - 1;
    int goff[UNR], koff[UNR], voff[UNR];          // global offset (floats), LDS byte offsets; -1: no unit
    // For head dims of 32 / 64 / 128 chunks (DCH 1 / 2 / 4) a row gets 8 DCH lanes (the last ones idle when hd is not a multiple of
    // 32): the 16-lane groups of a ds_write_b64 then stay inside one row of either image and its banks (with 12 units per row dealt
    // to consecutive lanes, groups straddled two rows 128 bytes = 32 banks apart: 0.7 conflict cycles per LDS instruction).
    constexpr bool ROWGROUPS = DCH != 3;
    constexpr int LPR = 8 * DCH, RPP = T / LPR;   // lanes per row, rows per unit index u
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const int idx = u * T + tid;
        const int row = ROWGROUPS ? u * RPP + tid / LPR : (int)(((float)idx + 0.5f) * inv_dgr);     // exact: idx < 2^14
        const int d4 = ROWGROUPS ? tid % LPR : idx - row * dgr;
        const bool ok = ROWGROUPS ? (row < N && d4 < dgr) : idx < units;
        const int sw = SPK == 16 ? (row & 15) : SPK == 8 ? ((row >> 1) & 7) : ((0x1320 >> (((row >> 2) & 3) * 4)) & 3);
        goff[u] = ok ? row * C3 + (d4 << 2) : 0;
        koff[u] = ok ? row * KS + ((((d4 >> 1) ^ sw) & ksw_mask) << 4) + ((d4 & 1) << 3) : -1;
        voff[u] = row * VS + (d4 << 3);
    }

    auto item_of = [&](int v, int& b, int& h) {
        const int slot = v >> 3;
        b = (slot / heads) * 8 + (v & 7);
        h = slot % heads;
    };
    auto convert_store = [&](const f32x4 k4, const f32x4 v4, int ko, int vo) {
        f16x2 h0, l0, h1, l1;
        split_pair((f32x2){k4.x, k4.y}, h0, l0);
        split_pair((f32x2){k4.z, k4.w}, h1, l1);
        *reinterpret_cast<f16x4*>(Kh + ko) = (f16x4){h0.x, h0.y, h1.x, h1.y};
        *reinterpret_cast<f16x4*>(Kl + ko) = (f16x4){l0.x, l0.y, l1.x, l1.y};
        split_pair((f32x2){v4.x, v4.y}, h0, l0);
        split_pair((f32x2){v4.z, v4.w}, h1, l1);
        *reinterpret_cast<f16x4*>(Vh + vo) = (f16x4){h0.x, h0.y, h1.x, h1.y};
        *reinterpret_cast<f16x4*>(Vl + vo) = (f16x4){l0.x, l0.y, l1.x, l1.y};
    };

    int v = blockIdx.x, b, h;
    item_of(v, b, h);
    while (v < vitems && b >= Bw) {               // the padded tail of the virtual item space (Bw not a multiple of 8)
        v += gridDim.x;
        item_of(v, b, h);
    }
    if (v >= vitems) return;

    f32x4 k0[UNR], v0[UNR];                       // the rows of the current item (requested one item ahead)
    auto request0 = [&](int bb, int hh) {
        const float* kvbase = qkv + (long long)((bb + kv_shift) % Bw) * N * C3 + C + hh * hd;
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (ROWGROUPS ? u * RPP < N : u * T < units) {      // wave-uniform: the units past the window's rows are nobody's
                k0[u] = *reinterpret_cast<const f32x4*>(kvbase + goff[u]);
                v0[u] = *reinterpret_cast<const f32x4*>(kvbase + goff[u] + C);
            }
        }
    };
    request0(b, h);
    // Q fragments of this wave (B operand: lane holds d = 32 c + 8 g .. + 7 of its query), also requested one item ahead: right
    // after the QK^T product, into the registers its operands leave.  Requested at the top of an item they would sit BEHIND the
    // previous item's stores in the in-order vmcnt, and every item would wait for a store round trip plus a load round trip
    // (SQ_WAIT_ANY was 65 % of the wave cycles).
    f32x4 qx[DCH][2];
    auto request_q = [&](int bb, int hh) {
        const float* qrow = qkv + ((long long)bb * N + (qok ? q : 0)) * C3 + hh * hd;
#pragma unroll
        for (int c = 0; c < DCH; ++c)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int d = 32 * c + 8 * g + 4 * hf;
                qx[c][hf] = *reinterpret_cast<const f32x4*>(qrow + (d < hd ? d : 0));
            }
    };
    request_q(b, h);
#ifdef ATMVFI_STAMP
    const bool stamping = g_attn_stamp && blockIdx.x == 0 && tid < 64;
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0, titems = 0;
#endif

    const int rsw = SPK == 16 ? r : SPK == 8 ? (r >> 1) : ((0x1320 >> ((r >> 2) * 4)) & 3);     // the swizzle of rows 16 kt + r
    const unsigned vlane = (unsigned)((4 * g + ((lane >> 2) & 3)) * VS + ((lane & 3) << 3));
    const unsigned vh0 = lds_offset(Vh) + vlane, vl0 = lds_offset(Vl) + vlane;
    const float qy = floorf(((float)q + 0.5f) * inv_ws), qxc = (float)q - qy * (float)ws;
    constexpr float LOG2E = 1.4426950408889634f;
    const float sl2 = scale * LOG2E;

    for (;;) {
#ifdef ATMVFI_STAMP
        if (stamping) { tlast = __builtin_amdgcn_s_memtime(); ++titems; }
#endif
        // ---- stage K and V (fp32 -> fp16 hi / lo') from the registers ----
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (koff[u] >= 0) convert_store(k0[u], v0[u], koff[u], voff[u]);
        AT_STAMP(0);
        if (labels && tid < N) Ls[tid] = labels[(long long)(b % nW) * N + tid];      // 64 NT threads >= N

        // split the Q fragments (zero outside the head dim / the window)
        f16x8_t qh[DCH], ql[DCH];
#pragma unroll
        for (int c = 0; c < DCH; ++c) {
            if (c < D32) {
                f16x2 hh[4], ll[4];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const bool ok = qok && (32 * c + 8 * g + 4 * hf) < hd;
                    const f32x4 x = qx[c][hf];
                    split_pair(ok ? (f32x2){x.x, x.y} : (f32x2){0.f, 0.f}, hh[2 * hf], ll[2 * hf]);
                    split_pair(ok ? (f32x2){x.z, x.w} : (f32x2){0.f, 0.f}, hh[2 * hf + 1], ll[2 * hf + 1]);
                }
                qh[c] = (f16x8_t){hh[0].x, hh[0].y, hh[1].x, hh[1].y, hh[2].x, hh[2].y, hh[3].x, hh[3].y};
                ql[c] = (f16x8_t){ll[0].x, ll[0].y, ll[1].x, ll[1].y, ll[2].x, ll[2].y, ll[3].x, ll[3].y};
            }
        }
        AT_STAMP(1);
        __syncthreads();
        AT_STAMP(2);

        // ---- the next item of this workgroup; its first rows are on their way while this item is computed ----
        int vn = vitems, bn = 0, hn = 0;
        if constexpr (PERSIST) {
            vn = v + gridDim.x;
            item_of(vn, bn, hn);
            while (vn < vitems && bn >= Bw) {
                vn += gridDim.x;
                item_of(vn, bn, hn);
            }
        }
        const bool has_next = PERSIST && vn < vitems;
        if (PREFETCH && has_next) request0(bn, hn);
        AT_STAMP(3);

        // ---- S^T = K Q^T ----
        f32x4 s[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, cor = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < DCH; ++c) {
                if (c < D32) {
                    const int off = (16 * kt + r) * KS + ((((4 * c + g) ^ rsw) & ksw_mask) << 4);
                    const f16x8_t kh = *reinterpret_cast<const f16x8_t*>(Kh + off);
                    const f16x8_t kl = *reinterpret_cast<const f16x8_t*>(Kl + off);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, qh[c], acc, 0, 0, 0);
                    cor = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, ql[c], cor, 0, 0, 0);
                    cor = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, qh[c], cor, 0, 0, 0);
                }
            }
            s[kt] = acc + cor * (1.0f / 1024.0f);
        }

        if (PREFETCH_Q && has_next) request_q(bn, hn);
        AT_STAMP(4);
        // ---- scale + mask + softmax numerators over keys (attention.py:192-200) in base 2: exp(x - m) = 2^((x - m) log2 e); the
        // division by the sum is applied to the outputs (a lane owns ONE query), so P stays the numerators in (0, 1] ----
        float mx = -INFINITY;
        if (labels) {
            const int lab_q = Ls[qok ? q : 0];
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                const int4 lk = *reinterpret_cast<const int4*>(Ls + 16 * kt + 4 * g);     // padded keys: stale, masked below
                const int lks[4] = {lk.x, lk.y, lk.z, lk.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = s[kt][e] * sl2;
                    x += lks[e] != lab_q ? -100.0f * LOG2E : 0.0f;
                    s[kt][e] = x;
                }
            }
        } else {
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) s[kt] *= sl2;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)                // only the last tile has padded keys (NT = ceil(N / 16))
            if (16 * (NT - 1) + 4 * g + e >= N) s[NT - 1][e] = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = fmaxf(mx, s[kt][e]);
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float p = __builtin_amdgcn_exp2f(s[kt][e] - mx);
                s[kt][e] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;

        // ---- P^T fragments: tiles 2 kt2 and 2 kt2 + 1 of the S registers, split in place ----
        f16x8_t ph[KT2], pl[KT2];
#pragma unroll
        for (int k2 = 0; k2 < KT2; ++k2) {
            const f32x4 a = s[2 * k2];
            const f32x4 c = (2 * k2 + 1 < NT) ? s[2 * k2 + 1 < NT ? 2 * k2 + 1 : 0] : (f32x4){0.f, 0.f, 0.f, 0.f};
            f16x2 hh[4], ll[4];
            split_pair((f32x2){a.x, a.y}, hh[0], ll[0]);
            split_pair((f32x2){a.z, a.w}, hh[1], ll[1]);
            split_pair((f32x2){c.x, c.y}, hh[2], ll[2]);
            split_pair((f32x2){c.z, c.w}, hh[3], ll[3]);
            ph[k2] = (f16x8_t){hh[0].x, hh[0].y, hh[1].x, hh[1].y, hh[2].x, hh[2].y, hh[3].x, hh[3].y};
            pl[k2] = (f16x8_t){ll[0].x, ll[0].y, ll[1].x, ll[1].y, ll[2].x, ll[2].y, ll[3].x, ll[3].y};
        }

        AT_STAMP(5);
        // ---- O^T = V^T P^T.  Transposed read: lane 4 qq + p of a 16-lane group supplies the address of block row qq, columns
        // 4 p .. 4 p + 3; lane i receives column i of the 4 rows.  EXEC is all ones here (every lane of the wave walks the items). ----
        const long long orow = (long long)b * N + (qok ? q : 0);
        for (int dt = 0; dt < DT; ++dt) {
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, cor = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k2 = 0; k2 < KT2; ++k2) {
                const unsigned o0 = (unsigned)(32 * k2 * VS + 32 * dt), o1 = o0 + (unsigned)(16 * VS);
                const f16x8_t vh = tr_pair(vh0 + o0, vh0 + o1), vl = tr_pair(vl0 + o0, vl0 + o1);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph[k2], acc, 0, 0, 0);
                cor = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl[k2], cor, 0, 0, 0);
                cor = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph[k2], cor, 0, 0, 0);
            }
            const f32x4 o = (acc + cor * (1.0f / 1024.0f)) * inv;
            const int d = 16 * dt + 4 * g;
            if (qok && d < hd) sink_store4(out, orow, h * hd + d, o);
            if (motion && qok && d == hd) {       // sum_k P (k_xy - q_xy) = sum_k P k_xy - q_xy   (sum_k P = 1)
                float* mp = motion + (((long long)b * N + q) * heads + h) * 2;
                mp[0] = o[0] - qxc;
                mp[1] = o[1] - qy;
            }
        }
        AT_STAMP(6);
        if (!has_next) break;
        v = vn;
        b = bn;
        h = hn;
        if (!PREFETCH) request0(b, h);
        if (!PREFETCH_Q) request_q(b, h);
        __syncthreads();                          // every wave is done with the images
        AT_STAMP(7);
    }
#ifdef ATMVFI_STAMP
    if (stamping && tid == 0) {
        for (int i = 0; i < 8; ++i) g_attn_stamp[i] = tacc[i];
        g_attn_stamp[8] = titems;
    }
#endif
}

__global__ void motion_head_kernel(const float* __restrict__ motion, const int* __restrict__ row_map,
                                   const float* __restrict__ w0, const float* __restrict__ b0,
                                   const float* __restrict__ w1, const float* __restrict__ b1,
                                   float* __restrict__ out, int out_ld, long long out_gstride, int out_rpg,
                                   long long rows, int heads, _Float16* __restrict__ phi, _Float16* __restrict__ plo,
                                   long long plane_rows, int pc0, int pgc) {
    fp16_saturate_on();
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= rows) return;
    const long long ro = row_map ? row_map[m] : m;
    if (ro < 0) return;
    const float* mp = motion + m * heads * 2;
    const int hid = heads >> 1;
    float ox = b1[0], oy = b1[0];
    for (int j = 0; j < hid; ++j) {
        float ax = b0[j], ay = b0[j];
        for (int hh = 0; hh < heads; ++hh) {
            const float wv = w0[j * heads + hh];
            ax += wv * mp[hh * 2];
            ay += wv * mp[hh * 2 + 1];
        }
        ox += w1[j] * gelu_erf(ax);
        oy += w1[j] * gelu_erf(ay);
    }
    const long long off = (out_rpg > 0) ? (ro / out_rpg) * out_gstride + (ro % out_rpg) * (long long)out_ld
                                        : ro * (long long)out_ld;
    out[off] = ox;
    out[off + 1] = oy;
    if (phi) {
        // the same two values again as split planes (the motion channels of the motion MLP's plane input: no separate split pass)
        const long long grp = out_rpg > 0 ? ro / out_rpg : 0;
        const long long prow = out_rpg > 0 ? ro - grp * out_rpg : ro;
        const int c = pc0 + (int)grp * pgc;
        f16x2 h, l;
        split_pair((f32x2){ox, oy}, h, l);
        const long long poff = ((long long)(c >> 5) * plane_rows + prow) * 32 + (c & 31);
        *reinterpret_cast<f16x2*>(phi + poff) = h;
        *reinterpret_cast<f16x2*>(plo + poff) = l;
    }
}

template <int NT>
int launch_attn(const float* qkv, const RowSink out, float* motion, const int* labels, int Bw, int nW, int N, int ws,
                int heads, int hd, int kv_shift, hipStream_t s) {
    const int dpad = (hd + 15) / 16 * 16, npad = NT * 16;
    const size_t lds = ((size_t)npad * ((dpad / 4 + 15) / 16 * 16) * 4 + (size_t)dpad * ((npad / 4 + 15) / 16 * 16) * 4 + npad) * sizeof(float);
    ATMVFI_REQUIRE(hd <= (NT >= 13 ? 64 : 128), ATMVFI_EINVAL, "window_attention: head dim %d too large for a %d-token window", hd, N);
    ATMVFI_REQUIRE(lds <= 160 * 1024, ATMVFI_EINVAL,
                   "window_attention: K/V tile of %zu bytes exceeds the 160 KiB LDS (ws %d, hd %d)", lds, ws, hd);
    auto kern = window_attn_kernel<NT>;
    if (lds > 48 * 1024) {       // raised once per device and kernel instance (common.h), not on every launch
        const hipError_t e = atmvfi::allow_dynamic_lds<window_attn_kernel<NT>>(lds);
        ATMVFI_REQUIRE(e == hipSuccess, ATMVFI_ELAUNCH, "window_attention: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    const float scale = 1.0f / sqrtf((float)hd);
    hipLaunchKernelGGL(kern, dim3((unsigned)((Bw + 7) / 8 * 8 * heads)), dim3(64 * NT), lds, s, qkv, out, motion, labels, N, nW, ws,
                       heads, hd, heads * hd, Bw, kv_shift, scale);
    return atmvfi::check_launch("window_attention");
}

template <int NT, int DCH>
int launch_attn_x3(const float* qkv, const RowSink out, float* motion, const int* labels, int Bw, int nW, int N, int ws,
                   int heads, int hd, int kv_shift, hipStream_t s) {
    const int d32 = (hd + 31) / 32, dt = (hd + (motion ? 4 : 0) + 15) / 16, npad = NT * 16, vrows = 32 * ((NT + 1) / 2);
    const int ks = (d32 <= 1 ? 4 : d32 == 2 ? 8 : 16) * 16, vs = 32 * dt + ((dt & 1) ? 0 : 32);      // as in the kernel
    const size_t lds = 2 * (size_t)npad * ks + 2 * (size_t)vrows * vs + (size_t)npad * sizeof(int);
    ATMVFI_REQUIRE(hd <= (NT >= 13 ? 64 : 128), ATMVFI_EINVAL, "window_attention: head dim %d too large for a %d-token window", hd, N);
    ATMVFI_REQUIRE(lds <= 160 * 1024, ATMVFI_EINVAL,
                   "window_attention: K/V tile of %zu bytes exceeds the 160 KiB LDS (ws %d, hd %d)", lds, ws, hd);
    if (lds > 48 * 1024) {
        const hipError_t e = atmvfi::allow_dynamic_lds<window_attn_x3_kernel<NT, DCH>>(lds);
        ATMVFI_REQUIRE(e == hipSuccess, ATMVFI_ELAUNCH, "window_attention: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    const float scale = 1.0f / sqrtf((float)hd);
    // persistent grid: as many workgroups as the CUs hold at once (LDS, registers, threads: asked of the runtime once per instance
    // and LDS size), a multiple of 8, at most one per item
    const int vitems = (Bw + 7) / 8 * 8 * heads;
    // (lds bytes << 8) | workgroups per CU, cached PER DEVICE like common.h's cu_count / allow_dynamic_lds: one process may drive
    // several GPUs (the occupancy answer belongs to the current device's copy of the function)
    static std::atomic<long long> occ_cache[atmvfi::kMaxDevices];
    int dev = 0;
    const bool cached = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < atmvfi::kMaxDevices;
    long long oc = cached ? occ_cache[dev].load(std::memory_order_relaxed) : 0;
    if ((oc >> 8) != (long long)lds || (oc & 255) == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, window_attn_x3_kernel<NT, DCH>, 64 * NT, lds) != hipSuccess || n < 1) n = 1;
        oc = ((long long)lds << 8) | (n > 255 ? 255 : n);
        if (cached) occ_cache[dev].store(oc, std::memory_order_relaxed);
    }
    const int per_cu = (int)(oc & 255);
    int grid = atmvfi::cu_count() * per_cu;
    if (grid > vitems || NT > 8) grid = vitems;   // vitems is a multiple of 8; NT > 8: one item per workgroup (PERSIST in the kernel)
    hipLaunchKernelGGL((window_attn_x3_kernel<NT, DCH>), dim3((unsigned)grid), dim3(64 * NT), lds, s, qkv, out, motion, labels, N, nW, ws,
                       heads, hd, heads * hd, Bw, kv_shift, scale, vitems);
    return atmvfi::check_launch("window_attention_f16x3");
}

template <int NT>
int launch_attn_x3_d(const float* qkv, const RowSink out, float* motion, const int* labels, int Bw, int nW, int N, int ws,
                     int heads, int hd, int kv_shift, hipStream_t s) {
    ATMVFI_REQUIRE(hd <= (NT >= 13 ? 64 : 128), ATMVFI_EINVAL, "window_attention: head dim %d too large for a %d-token window", hd, N);
    switch ((hd + 31) / 32) {
        case 1: return launch_attn_x3<NT, 1>(qkv, out, motion, labels, Bw, nW, N, ws, heads, hd, kv_shift, s);
        case 2: return launch_attn_x3<NT, 2>(qkv, out, motion, labels, Bw, nW, N, ws, heads, hd, kv_shift, s);
    }
    if constexpr (NT <= 12) {
        if (hd <= 96) return launch_attn_x3<NT, 3>(qkv, out, motion, labels, Bw, nW, N, ws, heads, hd, kv_shift, s);
        return launch_attn_x3<NT, 4>(qkv, out, motion, labels, Bw, nW, N, ws, heads, hd, kv_shift, s);
    }
    return ATMVFI_EINVAL;
}

int attention_entry(bool x3, const float* qkv, float* out, float* motion, const int32_t* labels, int Bw, int nW, int ws, int heads,
                    int hd, int kv_shift, void* out_hi, void* out_lo, int plane_ld, void* stream) {
    ATMVFI_REQUIRE(qkv, ATMVFI_EINVAL, "window_attention: null pointer");
    ATMVFI_REQUIRE(sink_ok(out, heads * hd, heads * hd, out_hi, out_lo, plane_ld, (long long)Bw * ws * ws), ATMVFI_EALIGN,
                   "window_attention: output needs fp32 rows and/or both fp16 planes (plane rows >= Bw*ws*ws), 16-byte aligned");
    const RowSink sink{out, heads * hd, (_Float16*)out_hi, (_Float16*)out_lo, plane_ld};
    ATMVFI_REQUIRE(Bw > 0 && nW > 0 && Bw % nW == 0, ATMVFI_EINVAL, "window_attention: Bw %d must be a positive multiple of nW %d", Bw, nW);
    ATMVFI_REQUIRE(ws >= 1 && ws <= 16, ATMVFI_EINVAL, "window_attention: window size %d outside 1..16", ws);
    ATMVFI_REQUIRE(heads > 0 && hd > 0 && hd % 4 == 0 && hd <= 128, ATMVFI_EINVAL, "window_attention: head dim %d must be a multiple of 4, at most 128", hd);
    ATMVFI_REQUIRE(kv_shift >= 0 && kv_shift < Bw, ATMVFI_EINVAL, "window_attention: kv_shift out of range");
    ATMVFI_REQUIRE(atmvfi::aligned16(qkv), ATMVFI_EALIGN, "window_attention: qkv must be 16-byte aligned");
    ATMVFI_REQUIRE(((long long)Bw + 8) * heads < (1ll << 31), ATMVFI_EINVAL, "window_attention: grid too large");
    const int N = ws * ws;
    const int nt = (N + 15) / 16;
    hipStream_t s = (hipStream_t)stream;
#define ATMVFI_ATTN_CASE(k)                                                                                              \
    case k:                                                                                                              \
        return x3 ? launch_attn_x3_d<k>(qkv, sink, motion, labels, Bw, nW, N, ws, heads, hd, kv_shift, s)                 \
                  : launch_attn<k>(qkv, sink, motion, labels, Bw, nW, N, ws, heads, hd, kv_shift, s);
    switch (nt) {
        ATMVFI_ATTN_CASE(1) ATMVFI_ATTN_CASE(2) ATMVFI_ATTN_CASE(3) ATMVFI_ATTN_CASE(4) ATMVFI_ATTN_CASE(5)
        ATMVFI_ATTN_CASE(6) ATMVFI_ATTN_CASE(7) ATMVFI_ATTN_CASE(8) ATMVFI_ATTN_CASE(9) ATMVFI_ATTN_CASE(10)
        ATMVFI_ATTN_CASE(11) ATMVFI_ATTN_CASE(12) ATMVFI_ATTN_CASE(13) ATMVFI_ATTN_CASE(14) ATMVFI_ATTN_CASE(15)
        ATMVFI_ATTN_CASE(16)
    }
#undef ATMVFI_ATTN_CASE
    atmvfi::set_error("window_attention: unsupported token count %d", N);
    return ATMVFI_EINVAL;
}

}  // namespace

#ifdef ATMVFI_STAMP
extern "C" int atmvfi_debug_set_attn_stamp_buffer(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamp), &buf, sizeof(buf)) == hipSuccess ? 0 : ATMVFI_ELAUNCH;
}
#endif

extern "C" int atmvfi_window_attention_f16x3(const float* qkv, float* out, float* motion, const int32_t* labels, int Bw,
                                              int nW, int ws, int heads, int hd, int kv_shift, void* out_hi, void* out_lo,
                                              int plane_ld, void* stream) {
    return attention_entry(true, qkv, out, motion, labels, Bw, nW, ws, heads, hd, kv_shift, out_hi, out_lo, plane_ld, stream);
}

extern "C" int atmvfi_window_attention(const float* qkv, float* out, float* motion, const int32_t* labels, int Bw,
                                        int nW, int ws, int heads, int hd, int kv_shift, void* out_hi, void* out_lo,
                                        int plane_ld, void* stream) {
    return attention_entry(false, qkv, out, motion, labels, Bw, nW, ws, heads, hd, kv_shift, out_hi, out_lo, plane_ld, stream);
}

extern "C" int atmvfi_window_attn_cross_motion(const float* qkv, float* out, float* motion, const int32_t* labels,
                                                int Bw, int nW, int ws, int heads, int hd, void* stream) {
    ATMVFI_REQUIRE(motion, ATMVFI_EINVAL, "window_attn_cross_motion: motion output required");
    ATMVFI_REQUIRE(Bw % 2 == 0, ATMVFI_EINVAL, "window_attn_cross_motion: Bw must be even (two frames)");
    return atmvfi_window_attention(qkv, out, motion, labels, Bw, nW, ws, heads, hd, Bw / 2, nullptr, nullptr, 0, stream);
}

extern "C" int atmvfi_window_attn_self(const float* qkv, float* out, const int32_t* labels, int Bw, int nW, int ws,
                                        int heads, int hd, void* stream) {
    return atmvfi_window_attention(qkv, out, nullptr, labels, Bw, nW, ws, heads, hd, 0, nullptr, nullptr, 0, stream);
}

extern "C" int atmvfi_motion_head_planes(const float* motion, const int32_t* row_map, const float* w0, const float* b0,
                                         const float* w1, const float* b1, float* out, int out_ld, int64_t out_gstride, int out_rpg,
                                         int64_t rows, int heads, void* out_hi, void* out_lo, int64_t plane_rows, int plane_c0, int plane_gc,
                                         void* stream) {
    ATMVFI_REQUIRE(motion && w0 && b0 && w1 && b1 && out, ATMVFI_EINVAL, "motion_head: null pointer");
    ATMVFI_REQUIRE(rows > 0 && heads >= 2 && heads % 2 == 0, ATMVFI_EINVAL, "motion_head: bad rows/heads");
    ATMVFI_REQUIRE((out_hi == nullptr) == (out_lo == nullptr), ATMVFI_EINVAL, "motion_head: the plane sink needs both planes");
    if (out_hi)
        ATMVFI_REQUIRE(plane_rows > 0 && plane_c0 >= 0 && plane_c0 % 2 == 0 && plane_gc % 2 == 0 && atmvfi::aligned16(out_hi) && atmvfi::aligned16(out_lo),
                       ATMVFI_EINVAL, "motion_head: plane sink needs plane_rows > 0, even channel offsets and 16-byte aligned planes");
    const unsigned blocks = (unsigned)((rows + 255) / 256);
    hipLaunchKernelGGL(motion_head_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, motion, row_map, w0, b0, w1,
                       b1, out, out_ld, (long long)out_gstride, out_rpg, (long long)rows, heads, (_Float16*)out_hi, (_Float16*)out_lo,
                       (long long)plane_rows, plane_c0, plane_gc);
    return atmvfi::check_launch("motion_head");
}

extern "C" int atmvfi_motion_head(const float* motion, const int32_t* row_map, const float* w0, const float* b0, const float* w1,
                                  const float* b1, float* out, int out_ld, int64_t out_gstride, int out_rpg, int64_t rows, int heads,
                                  void* stream) {
    return atmvfi_motion_head_planes(motion, row_map, w0, b0, w1, b1, out, out_ld, out_gstride, out_rpg, rows, heads, nullptr, nullptr, 0, 0,
                                     0, stream);
}

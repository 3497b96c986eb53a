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

__global__ void motion_head_kernel(const float* __restrict__ motion, const int* __restrict__ row_map,
                                   const float* __restrict__ w0, const float* __restrict__ b0,
                                   const float* __restrict__ w1, const float* __restrict__ b1,
                                   float* __restrict__ out, int out_ld, long long out_gstride, int out_rpg,
                                   long long rows, int heads) {
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

}  // namespace

extern "C" int atmvfi_window_attention(const float* qkv, float* out, float* motion, const int32_t* labels, int Bw,
                                        int nW, int ws, int heads, int hd, int kv_shift, void* out_hi, void* out_lo,
                                        int plane_ld, void* stream) {
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
#define ATMVFI_ATTN_CASE(k) \
    case k: return launch_attn<k>(qkv, sink, motion, labels, Bw, nW, N, ws, heads, hd, kv_shift, s);
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

extern "C" int atmvfi_motion_head(const float* motion, const int32_t* row_map, const float* w0, const float* b0,
                                   const float* w1, const float* b1, float* out, int out_ld, int64_t out_gstride,
                                   int out_rpg, int64_t rows, int heads, void* stream) {
    ATMVFI_REQUIRE(motion && w0 && b0 && w1 && b1 && out, ATMVFI_EINVAL, "motion_head: null pointer");
    ATMVFI_REQUIRE(rows > 0 && heads >= 2 && heads % 2 == 0, ATMVFI_EINVAL, "motion_head: bad rows/heads");
    const unsigned blocks = (unsigned)((rows + 255) / 256);
    hipLaunchKernelGGL(motion_head_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, motion, row_map, w0, b0, w1,
                       b1, out, out_ld, (long long)out_gstride, out_rpg, (long long)rows, heads);
    return atmvfi::check_launch("motion_head");
}

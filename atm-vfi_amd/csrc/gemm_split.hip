// Split-plane GEMM for gfx950: the "f16x3" arithmetic of gemm_f16x3.hip on activations that are ALREADY stored as
// two fp16 planes (hi, lo' = (x - hi) * 1024), so the operand path is pure LDS-DMA:
//   global_load_lds_dwordx4 (16 B per lane, no VGPR destination, no VALU split) -> 3-stage LDS ring ->
//   ds_read_b128 fragments -> v_mfma_f32_16x16x32_f16.
// The fp32-input engine splits every activation tile once per COLUMN block (12x for a 1536-wide layer) on the
// VALU and stages it through registers; here the producer kernel splits once, in its epilogue.
//
// Block = 512 threads = 8 wavefronts as 4 (rows) x 2 (columns); each wave owns 64 rows x 64 columns
// (4 x 4 MFMA tiles, acc + cor), block tile BM = 256 rows x BN = 128 columns, K in steps of 32.
// One LDS stage = A_hi[256][32] A_lo[256][32] W_hi[128][32] W_lo[128][32] halves = 48 KiB; three stages.
// LDS rows are 64 B; the slot swizzle ((row>>2)&1)<<1 is applied on the SOURCE address of the DMA (its
// destination is lane-linear) and again on the fragment read.
// Both operands are stored K-STEP MAJOR in global memory -- activations [32-channel chunk][row][32] (RowSink, common.h), weights
// [k-step][row][32] (atmvfi_pack_weight_split) -- so the 16 rows x 64 B of one DMA instruction are one contiguous KiB.
// Per k-step: s_waitcnt vmcnt(6) (own pieces of this stage landed) -> s_barrier (everyone's landed, everyone is
// done with the previous stage) -> DMA stage k+2 into the buffer of stage k-1 -> 16 ds_read_b128 + 48 MFMA.
#include "common.h"
#include "gemm_common.h"

// The kernel of THIS file is the REFERENCE SCHEDULE of the plane-input GEMM: the first form of the engine, kept as the yardstick the
// ping-pong kernels (gemm_pp.hip, gemm_duo.hip) are held to bit for bit (tests/test_gpu_ops.py::test_gemm_pp_matches_reference_schedule)
// and for the stamp / ablation tools.  The default forward never launches it, so the product library does not carry it (round 6):
// it is compiled only into the diagnostic libraries (`make ref`, `make stamp`, `make ablate` -> tools/lib/).  What the product
// keeps from this file: the engine selection (launch_gemm_split), split_planes, head1x1_planes, the split-K plan.
#if defined(ATMVFI_REFSCHED) || defined(ATMVFI_STAMP) || defined(ATMVFI_ABLATE)
#define ATMVFI_HAVE_REFSCHED 1
#endif

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#ifdef ATMVFI_STAMP
// Diagnostic build only (`make stamp`, tools/stamp_split.py): shader-clock and wall-clock stamps around the k-loop.
static unsigned long long* g_split_stamp = nullptr;
extern "C" void atmvfi_debug_set_split_stamp_buffer(void* p) { g_split_stamp = (unsigned long long*)p; }
#define SPLIT_STAMP(i) do { if (a.stamp) { tstamp[i] = __builtin_amdgcn_s_memtime(); rstamp[i] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define SPLIT_STAMP(i) do { } while (0)
#endif

namespace {

using atmvfi::GemmDev;

constexpr float LO_UNSCALE = 1.0f / 1024.0f;

__device__ __forceinline__ int swz64(int row) { return ((row >> 2) & 1) << 1; }

typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ void dma16(const _Float16* src, _Float16* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// WGM x WGN wavefronts, each owning 64 rows x 64 columns (4 x 4 MFMA tiles, acc + cor); three LDS stages.
//
// Software pipeline (per wave, all register indices static):
//   * the four weight fragment pairs w[j] of a k-step stay in registers for the whole step;
//   * the activation fragment pairs x[i] sit in a 4-slot ring, fetched two "items" (12 MFMAs each) ahead of use;
//   * item i = { fetch x[i+2] ; 8 MFMAs (wl*xh, wh*xh) ; 4 MFMAs (wh*xl) }, and the last item refills w[j] for the
//     NEXT k-step right behind the last MFMA that reads it -- so no ds_read latency is exposed at a step boundary.
//   * fragments of stage k+1 are first read in item 2 of step k: the one barrier per step sits between items 1 and 2,
//     behind s_waitcnt vmcnt (own DMA pieces of stage k+1 landed) and lgkmcnt(0) (own reads of stage k complete),
//     and right after it the DMA for stage k+3 is issued into the buffer stage k just vacated (two steps of lookahead).
// Diagnostic build only (`make ablate`): ATMVFI_SPLIT_DEBUG bits 4 / 8 / 16 switch off the activation-fragment LDS reads of the
// loop, the activation DMA, the weight-fragment reloads (wrong results; prices the LDS traffic of each).
#ifdef ATMVFI_ABLATE
#define SABL(bit) ((a.dbg & (bit)) != 0)
#else
#define SABL(bit) false
#endif

#ifdef ATMVFI_HAVE_REFSCHED
template <int WGM, int WGN, bool CONVM>
__global__ __launch_bounds__(64 * WGM * WGN, 2) void gemm_split_kernel(const GemmDev a) {
    fp16_saturate_on();
    constexpr int NW = WGM * WGN;
    constexpr int NT = 64 * NW;
    constexpr int BM = 64 * WGM, BN = 64 * WGN;
    constexpr int STAGE_HALVES = (2 * BM + 2 * BN) * 32;
    constexpr int A_LO_OFF = BM * 32, W_HI_OFF = 2 * BM * 32;
    constexpr int APT = BM * 8 / NT, WPT = BN * 8 / NT;             // 16-byte DMA pieces per thread per stage
    constexpr int PIECES = APT + WPT;
    static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0, "pieces must divide evenly");
    static_assert(APT == 4 && WPT == 2, "issue_pair assumes 4 + 2 pieces per thread");
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 15;
    const int g = lane >> 4;
    const int wm = wave / WGN, wn = wave % WGN;

    // XCD-aware tile order: blocks b and b+8 share an XCD, so the column blocks of one row tile go to one L2
    const int xcd = blockIdx.x & 7;
    const int slot = blockIdx.x >> 3;
    const int mgrp = slot / a.nblocks;
    const int nblk = slot - mgrp * a.nblocks;
    const long long m0 = ((long long)mgrp * 8 + xcd) * BM;
    if (m0 >= a.M) return;
    const int n0 = nblk * BN;
    float* cst = reinterpret_cast<float*>(smem + 3 * STAGE_HALVES);       // epilogue constants of this column block (gemm_common.h)

    // ---- DMA pieces: P = (i*NW + wave)*64 + lane.  A: plane = P / (4*BM), row = (P>>2) % BM, physical slot = lane&3;
    //      the LDS image is linear in P (hi plane then lo plane), i.e. wave-uniform base + lane*16 B as LDS-DMA requires.
    const _Float16* asrc[APT];
    int adst[APT];
    // CONV mode (strided / dilated / 1x1 convolutions on split-plane input): k-step = (tap, 32-channel chunk); GEMM row m is output
    // pixel (n, oy, ox) and its operand row at tap (ky, kx) is input pixel (n, oy*stride - pad + ky*dil, ox*stride - pad + kx*dil),
    // or the planes' zero row N*H*W when that falls outside the image -- the same LDS-DMA with a per-lane source row.  Pieces i
    // and i + APT/2 are the hi / lo plane of the same rows.
    int ciy[APT / 2], cix[APT / 2];
    long long crow[APT / 2];
#pragma unroll
    for (int i = 0; i < APT; ++i) {
        const int P = (i * NW + wave) * 64 + lane;
        const int plane = P / (4 * BM);
        const int row = (P >> 2) % BM;
        const int ls = (lane & 3) ^ swz64(row);
        long long m = m0 + row;
        if (m >= a.M) m = a.M - 1;                                  // tail rows: valid address, result never stored
        asrc[i] = (plane ? a.a_lo : a.a_hi) + m * 32 + ls * 8;      // chunk kc adds kc * plane_rows * 32
        adst[i] = (i * NW + wave) * 64 * 8;
        if (CONVM && i < APT / 2) {
            const int hw = a.Ho * a.Wo;
            const int n = (int)(m / hw);
            const int rem = (int)(m - (long long)n * hw);
            const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
            ciy[i] = oy * a.stride - a.pad;
            cix[i] = ox * a.stride - a.pad;
            crow[i] = ((long long)n * a.H + ciy[i]) * a.W + cix[i];
        }
    }
    static_assert(APT == 4, "CONV mode assumes pieces 0,1 = hi plane and 2,3 = lo plane of the same rows");
    int c_tap = 0, c_chunk = 0;                                       // (tap, chunk) of the k-step whose pieces are issued next
    auto next_k = [&]() {
        if constexpr (CONVM) { if (++c_chunk == a.cpt32) { c_chunk = 0; ++c_tap; } }
    };
    const long long zero_row = (long long)a.in_N * a.H * a.W;
    auto a_src = [&](int i, int kc) -> const _Float16* {
        if constexpr (!CONVM) return asrc[i] + kc * ((long long)a.in_ld * 32);
        const int j = i & 1;
        const int ky = c_tap / a.kw, kx = c_tap - ky * a.kw;
        const int iy = ciy[j] + ky * a.dil, ix = cix[j] + kx * a.dil;
        const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const long long row = ok ? crow[j] + (long long)(ky * a.dil) * a.W + kx * a.dil : zero_row;
        const int ls8 = (((lane & 3) ^ swz64((((i * NW + wave) * 64 + lane) >> 2) % BM))) * 8;
        const bool second = c_chunk >= a.split_chunks;               // input channels from a second plane buffer (concat-free)
        const _Float16* base = second ? (i >= 2 ? a.a_lo2 : a.a_hi2) + (long long)(c_chunk - a.split_chunks) * a.in_ld2 * 32
                                      : (i >= 2 ? a.a_lo : a.a_hi) + (long long)c_chunk * a.in_ld * 32;
        return base + row * 32 + ls8;
    };
    const _Float16* wsrc[WPT];
    int wdst[WPT];
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int P = (i * NW + wave) * 64 + lane;
        const int plane = P / (4 * BN);
        const int row = (P >> 2) % BN;
        const int ls = (lane & 3) ^ swz64(row);
        int n = n0 + row;
        if (n >= a.wrows) n = a.wrows - 1;                          // columns past the packed rows: never stored
        wsrc[i] = (plane ? a.w_lo : a.w_hi) + (long long)n * 32 + ls * 8;       // k-step kc adds kc * wrows * 32
        wdst[i] = W_HI_OFF + (i * NW + wave) * 64 * 8;
    }
    const long long wstep = (long long)a.wrows * 32;      // halves per k-step of a weight plane
    auto issue = [&](int kc, int buf) {
        _Float16* st = smem + buf * STAGE_HALVES;
#pragma unroll
        for (int i = 0; i < APT; ++i) dma16(a_src(i, kc), st + adst[i]);
#pragma unroll
        for (int i = 0; i < WPT; ++i) dma16(wsrc[i] + kc * wstep, st + wdst[i]);
    };
    // one third of a stage's pieces (PIECES == 6: A pieces 0-3, W pieces 0-1)
    auto issue_pair = [&](int kc, int buf, int p) {
        _Float16* st = smem + buf * STAGE_HALVES;
        if (p == 0 && !SABL(8)) { dma16(a_src(0, kc), st + adst[0]); dma16(a_src(1, kc), st + adst[1]); }
        if (p == 1 && !SABL(8)) { dma16(a_src(2, kc), st + adst[2]); dma16(a_src(3, kc), st + adst[3]); }
        if (p == 2) { dma16(wsrc[0] + kc * wstep, st + wdst[0]); dma16(wsrc[1] + kc * wstep, st + wdst[1]); }
    };
    // The DMA of one stage is spread over three items (an LDS-DMA instruction holds the wave's issue for 60-180 cycles)
    // and the two wave groups of a SIMD (waves 0-3 / 4-7) take different items, so one group's MFMAs cover the other's
    // DMA issue.  slot: 0 = item 2 (right behind the barrier), 1 = item 3, 2 = item 0 of the next step, 3 = item 1.
    const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
    int ikc = 0x7fffffff, ibuf = 0;                                      // stage being issued (ikc >= nk: nothing)

    f32x4 acc[4][4], cor[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            cor[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }

    // fragment offsets (halves) inside a stage; swz64(16*i + row) == swz64(row), so tile i is a constant +16*32*i away
    const int xrow = 64 * wm + r, wrow = 64 * wn + r;
    const int xoff = xrow * 32 + ((g ^ swz64(xrow)) << 3);
    const int woff = W_HI_OFF + wrow * 32 + ((g ^ swz64(wrow)) << 3);
    f16x8 xh[4], xl[4], wh[4], wl[4];
    auto load_x = [&](const _Float16* st, int i) {
        xh[i] = *reinterpret_cast<const f16x8*>(st + xoff + i * 512);
        xl[i] = *reinterpret_cast<const f16x8*>(st + xoff + i * 512 + A_LO_OFF);
    };
    auto mma_main = [&](int i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            cor[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[j], xh[i], cor[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xh[i], acc[i][j], 0, 0, 0);
        }
    };
    auto mma_cross = [&](int i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) cor[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xl[i], cor[i][j], 0, 0, 0);
    };

#ifdef ATMVFI_STAMP
    unsigned long long tstamp[4], rstamp[4];
#endif
    SPLIT_STAMP(0);
    const int nk = (a.dbg & 2) ? 1 : a.nchunks32;
    atmvfi::gemm_dma_consts<BN>(a, n0, cst, wave, lane);      // oldest DMA of the wave: landed whenever stage 0 has
    issue(0, 0);
    next_k();
    if (nk > 1) issue(1, 1);
    next_k();
    if (nk > 2) issue(2, 2);
    next_k();                                   // (tap, chunk) of k-step 3, the first one issued inside the loop
    if (nk > 2) wait_vmcnt<2 * PIECES>();
    else if (nk > 1) wait_vmcnt<PIECES>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        wh[j] = *reinterpret_cast<const f16x8*>(smem + woff + j * 512);
        wl[j] = *reinterpret_cast<const f16x8*>(smem + woff + j * 512 + BN * 32);
    }
    load_x(smem, 0);
    load_x(smem, 1);

    int buf = 0;
    SPLIT_STAMP(1);
    for (int kc = 0; kc < nk; ++kc) {
        const _Float16* st = smem + buf * STAGE_HALVES;
        const int nbuf = buf == 2 ? 0 : buf + 1;
        const _Float16* sn = smem + nbuf * STAGE_HALVES;
        const bool more = kc + 1 < nk;
        // item 0
        if (!SABL(4)) load_x(st, 2);
        if (ikc < nk) issue_pair(ikc, ibuf, 2 - grp);
        mma_main(0);
        mma_cross(0);
        __builtin_amdgcn_sched_barrier(0);
        // item 1
        if (!SABL(4)) load_x(st, 3);
        if (ikc < nk && grp == 1) issue_pair(ikc, ibuf, 2);
        mma_main(1);
        mma_cross(1);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            // stage kc+1: own DMA pieces landed (stage kc+2 may still be in flight); own reads of stage kc complete
            if (kc + 2 < nk) wait_vmcnt<PIECES>();
            else wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (kc > 0) next_k();               // every A piece of stage kc + 2 has been issued (items 0, 1 above)
            ikc = kc + 3;
            ibuf = buf;
            if (!SABL(4)) load_x(sn, 0);
            if (ikc < nk && grp == 0) issue_pair(ikc, ibuf, 0);
        } else {
            ikc = 0x7fffffff;
        }
        // item 2
        mma_main(2);
        mma_cross(2);
        __builtin_amdgcn_sched_barrier(0);
        // item 3 (+ weight fragments of the next step, each right behind the last MFMA that reads the old one)
        if (more && !SABL(4)) load_x(sn, 1);
        if (ikc < nk) issue_pair(ikc, ibuf, 1 - grp);
        mma_main(3);
        if (more && !SABL(16)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) wl[j] = *reinterpret_cast<const f16x8*>(sn + woff + j * 512 + BN * 32);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            cor[3][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xl[3], cor[3][j], 0, 0, 0);
            if (more && !SABL(16)) wh[j] = *reinterpret_cast<const f16x8*>(sn + woff + j * 512);
        }
        __builtin_amdgcn_sched_barrier(0);
        buf = nbuf;
    }
    SPLIT_STAMP(2);

#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long m = m0 + 64 * wm + 16 * i + r;
        float* orow;
        const float* rrow;
        long long prow;
        int pc0;
        bool live = m < a.M && atmvfi::gemm_out_row(a, m, orow, rrow, prow, pc0);
        if (a.dbg & 1) live = live && acc[i][0].x == 12345.678f;
        // residual vectors of the whole row first (one wait), then the four stores back to back
        f32x4 res[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            res[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (a.residual && live) res[j] = atmvfi::gemm_load_residual4(rrow, atmvfi::gemm_chan_pos(a, n0 + 64 * wn + 16 * j + 4 * g));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cl = 64 * wn + 16 * j + 4 * g;
            const atmvfi::ChanPos cp = atmvfi::gemm_chan_pos(a, n0 + cl);
            const f32x4 b = *reinterpret_cast<const f32x4*>(cst + cl);
            const f32x4 p = *reinterpret_cast<const f32x4*>(cst + BN + cl);
            if (live) atmvfi::gemm_finish_store4(a, orow, prow, pc0, cp, acc[i][j] + cor[i][j] * LO_UNSCALE, b, p, res[j]);
        }
    }
#ifdef ATMVFI_STAMP
    SPLIT_STAMP(3);
    if (a.stamp && lane == 0) {
        unsigned long long* o = a.stamp + ((long long)blockIdx.x * NW + wave) * 8;
        for (int k = 0; k < 3; ++k) { o[k] = tstamp[k + 1] - tstamp[k]; o[4 + k] = rstamp[k + 1] - rstamp[k]; }
        o[3] = (unsigned long long)nk;
    }
#endif
}
#endif  // ATMVFI_HAVE_REFSCHED

// fp32 rows -> (hi, lo') planes in the chunk-major layout; one thread per 8 channels, pad channels of the last chunk written as
// zero.  An optional per-channel PReLU is applied first (the leading nn.PReLU of the decoder stages, network_base.py:209,215).
// The C channels go to plane channels c0 .. c0 + C (c0 a multiple of 8); channels up to the next multiple of `pad_to` (8 or 32)
// are written as zero.
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ in, int in_ld, long long M, int C,
                                                           const float* __restrict__ prelu, _Float16* hi, _Float16* lo, long long plane_rows,
                                                           int c0p, int pad_to) {
    fp16_saturate_on();
    const int groups = (C + pad_to - 1) / pad_to * (pad_to / 8);
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * groups) return;
    const long long m = t / groups;
    const int c0 = (int)(t - m * groups) * 8;
    float x[8];
    const float* p = in + m * in_ld + c0;
    if (c0 + 8 <= C && (in_ld & 3) == 0) {
        const f32x4 va = *reinterpret_cast<const f32x4*>(p), vb = *reinterpret_cast<const f32x4*>(p + 4);
        x[0] = va.x; x[1] = va.y; x[2] = va.z; x[3] = va.w; x[4] = vb.x; x[5] = vb.y; x[6] = vb.z; x[7] = vb.w;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = (c0 + e < C) ? p[e] : 0.f;
    }
    f16x8 h, l;
    if (prelu) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (c0 + e < C) x[e] = x[e] > 0.f ? x[e] : prelu[c0 + e] * x[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f16x2 hh, ll;
        split_pair((f32x2){x[2 * e], x[2 * e + 1]}, hh, ll);
        h[2 * e] = hh.x;
        h[2 * e + 1] = hh.y;
        l[2 * e] = ll.x;
        l[2 * e + 1] = ll.y;
    }
    const int cp = c0p + c0;
    const long long off = ((long long)(cp >> 5) * plane_rows + m) * 32 + (cp & 31);
    *reinterpret_cast<f16x8*>(hi + off) = h;
    *reinterpret_cast<f16x8*>(lo + off) = l;
}

// 1x1 convolution with at most 8 output channels on split-plane input: the read-out of a motion MLP (nn.Conv2d(hidden, 5, 1),
// network_base.py:158,195).  On the GEMM a 128-column tile is 96 % padding and the launch costs a whole tile's latency (0.03-0.05 ms
// for 50 MFLOP); here a lane owns a pixel row, reconstructs x = hi + lo' / 1024 (exact in fp32) from its 64-byte chunk rows --
// consecutive lanes read consecutive rows: 4 KiB per wave and chunk -- and accumulates against weights that arrive as scalar
// loads (wave-uniform).  fp32 FMAs on the full fp32 weights: at least as accurate as the f16x3 product it replaces.
template <int NO>
__global__ __launch_bounds__(256) void head1x1_planes_kernel(const _Float16* __restrict__ hi, const _Float16* __restrict__ lo, long long plane_rows,
                                                             long long rows, int cin, const float* __restrict__ w, const float* __restrict__ bias,
                                                             float* __restrict__ out, int out_ld, int cout) {
    fp16_saturate_on();
    // A block = 64 pixel rows x 4 waves; wave k takes the 32-channel chunks k, k + 4, ... (a lane per row with the whole K leaves 2
    // waves per CU walking 18-24 dependent chunk loads: 0.05 ms for 50 MFLOP), partial sums meet in LDS.  The weights of a wave's
    // chunk are wave-uniform: scalar loads.
    __shared__ float part[4][NO][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long row_raw = (long long)blockIdx.x * 64 + lane;
    const bool live = row_raw < rows;
    const long long row = live ? row_raw : rows - 1;
    float acc[NO];
#pragma unroll
    for (int n = 0; n < NO; ++n) acc[n] = 0.f;
    const int chunks = (cin + 31) >> 5;
    for (int ch = wave; ch < chunks; ch += 4) {
        const f16x8* ph = reinterpret_cast<const f16x8*>(hi + ((long long)ch * plane_rows + row) * 32);
        const f16x8* pl = reinterpret_cast<const f16x8*>(lo + ((long long)ch * plane_rows + row) * 32);
        f16x8 h[4], l[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) { h[v] = ph[v]; l[v] = pl[v]; }
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = ch * 32 + v * 8 + e;
                const int cw = c < cin ? c : cin - 1;            // pad channels of the last chunk are zero in the planes
                const float x = __builtin_fmaf((float)l[v][e], LO_UNSCALE, (float)h[v][e]);
#pragma unroll
                for (int n = 0; n < NO; ++n) acc[n] = __builtin_fmaf(w[(n < cout ? n : 0) * cin + cw], x, acc[n]);
            }
    }
#pragma unroll
    for (int n = 0; n < NO; ++n) part[wave][n][lane] = acc[n];
    __syncthreads();
    if (wave == 0 && live) {
#pragma unroll
        for (int n = 0; n < NO; ++n)
            if (n < cout) out[row * out_ld + n] = ((part[0][n][lane] + part[1][n][lane]) + (part[2][n][lane] + part[3][n][lane])) + (bias ? bias[n] : 0.f);
    }
}

#ifdef ATMVFI_HAVE_REFSCHED
template <int WGM, int WGN, bool CONVM>
int launch_split(const GemmDev& d, int ngemm, hipStream_t s) {
    constexpr int BM = 64 * WGM, BN = 64 * WGN;
    const size_t lds = (size_t)3 * (2 * BM + 2 * BN) * 32 * sizeof(_Float16) + atmvfi::gemm_const_floats(BN) * sizeof(float);
    auto kern = gemm_split_kernel<WGM, WGN, CONVM>;
    const hipError_t attr_err = atmvfi::allow_dynamic_lds<gemm_split_kernel<WGM, WGN, CONVM>>(lds);
    ATMVFI_REQUIRE(attr_err == hipSuccess, ATMVFI_ELAUNCH, "gemm_split: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    GemmDev dd = d;
#if defined(ATMVFI_ABLATE) || defined(ATMVFI_STAMP)
    static const int dbg = [] { const char* e = getenv("ATMVFI_SPLIT_DEBUG"); return e ? atoi(e) : 0; }();     // diagnostic builds only
    dd.dbg = dbg;
#else
    dd.dbg = 0;
#endif
#ifdef ATMVFI_STAMP
    dd.stamp = g_split_stamp;
#endif
    dd.nblocks = (ngemm + BN - 1) / BN;
    const long long mgroups = (atmvfi::ceil_div64(d.M, BM) + 7) / 8;
    ATMVFI_REQUIRE(mgroups * 8 * dd.nblocks < (1LL << 31), ATMVFI_EINVAL, "gemm_split: grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)(mgroups * 8 * dd.nblocks)), dim3(64 * WGM * WGN), lds, s, dd);
    return atmvfi::check_launch("gemm_split");
}
#endif  // ATMVFI_HAVE_REFSCHED

}  // namespace

// Split-K plan of an under-filled plane-input GEMM launch (gemm_duo.hip, 128 x 64 tiles, three workgroups per CU): as many K ranges as
// the idle workgroup slots take, each at least 8 k-steps, at most 8 ranges; taken when the model of tools/gemm_engines_ab.py
// (a 128 x 64 tile: 6.4 us + 0.6 us per k-step while the chip is under-filled, 0.75 with two and more per CU; the second kernel ~5 us)
// says at least 15 % less time.  Returns the number of ranges (1: no split).
static int gemm_splitk_plan(long long M, int ngemm, int nk) {
    const long long cus = atmvfi::cu_count();
    const long long tiles = atmvfi::ceil_div64(M, 128) * ((ngemm + 63) / 64);
    if (nk < 16 || tiles * 2 > cus) return 1;                      // only grids that leave at least half of the CUs without a tile
    const int S = (int)std::min<long long>(std::min<long long>(2 * cus / tiles, nk / 8), 8);     // at most two workgroups per CU
    if (S < 2) return 1;
    // 6.4 us + 0.6 us per k-step for a 128 x 64 tile on an under-filled chip (0.75 with more than one per CU); the second kernel ~5 us;
    // the partial sums go through memory once each way (~3 TB/s for this pattern) -- measured on the launches of network_lite
    // 256 x 256 (30 -> 16 us, 50 -> 17 us) and of network_base 576 x 960, where splitting the 68-374-tile launches LOST 8-25 us each
    const double per_step = tiles * S > cus ? 0.75 : 0.6;
    const double traffic_us = 2.0 * S * (double)M * atmvfi::round_up(ngemm, 64) * 4.0 / 3.0e6;
    const double t_unsplit = 6.4 + 0.6 * nk;
    const double t_split = 6.4 + per_step * ((nk + S - 1) / S) + 5.0 + traffic_us;
    return t_split < 0.85 * t_unsplit ? S : 1;
}

extern "C" int64_t atmvfi_gemm_workspace_floats(int64_t M, int ngemm, int ksteps) {
    if (M <= 0 || ngemm <= 0 || ksteps <= 0) return 0;
    const int S = gemm_splitk_plan(M, ngemm, ksteps);
    return S > 1 ? (int64_t)S * M * atmvfi::round_up(ngemm, 64) : 0;
}

int atmvfi::launch_gemm_split(const GemmDev& d, int ngemm, hipStream_t s) {
    // force_wn: -1 this file (the reference schedule: tools / A-B only), -2 / -4 gemm_duo.hip with 128- / 64-column tiles, -3 gemm_pp.hip;
    // 0 = choose.  The three give
    // bit-identical results, so the choice is a matter of time only.  Per launch (tools/duo_rule.py, both kernels timed inside a forward):
    //   * on a grid that fills the chip many times over the ping-pong kernel wins by 0-16 % (up to 57 % at K = 4608);
    //   * its persistent grid works in rounds of one 256 x 128 tile per CU, and a launch of 1.4 or 2.2 rounds pays for 2 or 3; the
    //     128 x 128 kernel's unit is half of that and two of them share a CU, so it wins whenever its own round count (+8 % for its
    //     lower per-tile rate, +15 % more for K >= 2048, + a fifth of a tile when a CU's last tile runs alone) comes out below;
    //   * layers of at most 64 columns always go to gemm_duo.hip, which runs them on 128 x 64 tiles (half of a 128-column tile would be
    //     padding: the refiner's 64 -> 64 stride-2 conv at 1080p 0.331 -> 0.248 ms, a 256 -> 64 conv at K = 2304 74 -> 52 us;
    //     tools/narrow_ab.py).
    // End to end: network_lite 256 x 256 +11 %, 256 x 448 +10 %, network_base 540p +3 %, 1080p +1 % (tools/small_gemm_ab.py).
    if (d.force_wn == -2 || d.force_wn == -4) return launch_gemm_duo(d, ngemm, s);
    if (d.force_wn == -3) return launch_gemm_pp(d, ngemm, s);
    if (d.force_wn == 0 && d.part) {
        const int S = gemm_splitk_plan(d.M, ngemm, d.nchunks32);
        const int ld = round_up(ngemm, 64);
        if (S > 1 && (long long)S * d.M * ld <= d.part_stride) {
            GemmDev ds = d;
            ds.ksplit = S;
            ds.part_ld = ld;
            ds.part_stride = d.M * ld;
            ds.force_wn = -4;
            return launch_gemm_duo(ds, ngemm, s);
        }
    }
    if (d.force_wn == 0) {
        if (ngemm <= 64) return launch_gemm_duo(d, ngemm, s);
        // Under-filled grids (round 4, tools/gemm_engines_ab.py -> profiles/r04_gemm_engines_ab.txt: all three schedules timed on every
        // launch of four configurations): while the 128 x 64 tiles of the launch number at most two per CU, the 128 x 64 form wins on
        // every launch measured -- its tile is a latency chain of 0.6 us per k-step + 6 us against 0.8 + 8 (128 x 128) and 1.0 + 9
        // (256 x 128), and the chip is not full either way.  network_lite 256 x 256: the 44 launches 0.775 -> 0.622 ms, 256 x 448
        // 0.553 -> 0.458, network_base 576 x 960 2.47 -> 2.35, 1088 x 1920 +-0.
        if (ceil_div64(d.M, 128) * ((ngemm + 63) / 64) <= 2 * (long long)cu_count()) {
            GemmDev d64 = d;
            d64.force_wn = -4;
            return launch_gemm_duo(d64, ngemm, s);
        }
        const long long cus = cu_count(), ntile = (ngemm + 127) / 128;
        const long long pp_rounds = ceil_div64(ceil_div64(d.M, 256) * ntile, cus);
        const long long duo_per_cu = ceil_div64(ceil_div64(d.M, 128) * ntile, cus);
        const double duo_rounds = (0.5 * (double)duo_per_cu + ((duo_per_cu & 1) ? 0.1 : 0.0)) * 1.08 * (d.nchunks32 >= 64 ? 1.15 : 1.0);
        return duo_rounds < (double)pp_rounds ? launch_gemm_duo(d, ngemm, s) : launch_gemm_pp(d, ngemm, s);
    }
    if (d.force_wn != -1) return launch_gemm_pp(d, ngemm, s);
#ifdef ATMVFI_HAVE_REFSCHED
    return d.mode == ATMVFI_GEMM_CONV ? launch_split<4, 2, true>(d, ngemm, s) : launch_split<4, 2, false>(d, ngemm, s);
#else
    ATMVFI_REQUIRE(false, ATMVFI_EINVAL, "gemm: tile_wn = -1 (the reference schedule) is not in the product library: load tools/lib/libatmvfi_hip_ref.so (make -C atm-vfi_amd/csrc ref)");
#endif
}

extern "C" int atmvfi_split_planes_at(const float* in, int in_ld, int64_t M, int C, const float* prelu, void* hi, void* lo, int plane_rows,
                                      int c0, int pad_to, void* stream) {
    ATMVFI_REQUIRE(in && hi && lo && M > 0 && C > 0, ATMVFI_EINVAL, "split_planes: bad arguments");
    ATMVFI_REQUIRE(plane_rows >= M && in_ld >= C, ATMVFI_EALIGN, "split_planes: plane rows %d must cover M and in_ld %d must cover C %d", plane_rows, in_ld, C);
    ATMVFI_REQUIRE(atmvfi::aligned16(hi) && atmvfi::aligned16(lo), ATMVFI_EALIGN, "split_planes: planes must be 16-byte aligned");
    ATMVFI_REQUIRE(c0 >= 0 && c0 % 8 == 0 && (pad_to == 8 || pad_to == 32) && (pad_to == 8 || c0 % 32 == 0), ATMVFI_EINVAL,
                   "split_planes: channel offset %d must be a multiple of 8 (of 32 when padding to 32), pad_to 8 or 32", c0);
    const long long n = (long long)M * ((C + pad_to - 1) / pad_to * (pad_to / 8));
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, in_ld, (long long)M, C,
                       prelu, (_Float16*)hi, (_Float16*)lo, (long long)plane_rows, c0, pad_to);
    return atmvfi::check_launch("split_planes");
}
extern "C" int atmvfi_split_planes(const float* in, int in_ld, int64_t M, int C, const float* prelu, void* hi, void* lo, int plane_rows,
                                   void* stream) {
    return atmvfi_split_planes_at(in, in_ld, M, C, prelu, hi, lo, plane_rows, 0, 32, stream);
}

extern "C" int atmvfi_head1x1_planes(const void* in_hi, const void* in_lo, int64_t plane_rows, int64_t rows, int Cin, const float* weight,
                                     const float* bias, int Cout, float* out, int out_ld, void* stream) {
    ATMVFI_REQUIRE(in_hi && in_lo && weight && out, ATMVFI_EINVAL, "head1x1_planes: null pointer");
    ATMVFI_REQUIRE(rows > 0 && plane_rows >= rows && Cin > 0 && Cout >= 1 && Cout <= 8 && out_ld >= Cout, ATMVFI_EINVAL,
                   "head1x1_planes: rows %lld (plane rows %lld), Cin %d, 1 <= Cout %d <= 8 <= ... out_ld %d", (long long)rows,
                   (long long)plane_rows, Cin, Cout, out_ld);
    ATMVFI_REQUIRE(atmvfi::aligned16(in_hi) && atmvfi::aligned16(in_lo), ATMVFI_EALIGN, "head1x1_planes: planes must be 16-byte aligned");
    const unsigned blocks = (unsigned)((rows + 63) / 64);
    if (Cout <= 5)
        hipLaunchKernelGGL(head1x1_planes_kernel<5>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)in_hi, (const _Float16*)in_lo,
                           (long long)plane_rows, (long long)rows, Cin, weight, bias, out, out_ld, Cout);
    else
        hipLaunchKernelGGL(head1x1_planes_kernel<8>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)in_hi, (const _Float16*)in_lo,
                           (long long)plane_rows, (long long)rows, Cin, weight, bias, out, out_ld, Cout);
    return atmvfi::check_launch("head1x1_planes");
}

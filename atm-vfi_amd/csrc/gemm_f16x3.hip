// Split-precision ("f16x3") variant of the implicit-GEMM engine for gfx950: nn.Linear rows,
// ConvTranspose2d(k2,s2) and the strided / dilated / 1x1 convolutions -- everything the 3x3
// stride-1 halo kernel (conv3x3_f16x3.hip) does not cover.  Same contract as gemm_conv.hip
// (modes, row groups, scatter map, in_prelu, epilogue), same arithmetic as conv3x3_f16x3:
//   x = hi + lo'/1024 (fp16 pair), hi*hi and hi*lo' + lo'*hi accumulated in two fp32
//   accumulators on v_mfma_f32_16x16x32_f16, folded with 2^-10 in the epilogue.
//
// Block = 512 threads = 8 wavefronts stacked along M: BM = 256 rows (two 16-row MFMA tiles per
// wave) x BN = 16*WN columns, K in chunks of 32 = (tap, 32 input channels).  fp32 activations
// are split while they are staged (registers -> two fp16 LDS planes); weights are pre-split.
// LDS rows are 64 bytes with the ((row>>2)&1)<<1 slot swizzle; operands of the NEXT chunk are
// fetched into registers before the MFMA phase and written to the other buffer after it.
#include "common.h"
#include "gemm_common.h"

#include <stdlib.h>


typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#ifdef ATMVFI_STAMP
// Diagnostic build only (`make stamp`, tools/stamp_gemm.py): per-wave tick sums of the phases of the persistent loop.
static unsigned long long* g_gemm_stamp = nullptr;
extern "C" void atmvfi_debug_set_gemm_stamp_buffer(void* p) { g_gemm_stamp = (unsigned long long*)p; }
#define GSTAMP(k) do { if (a.stamp) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tsum[k] += t_ - tlast; tlast = t_; } } while (0)
#else
#define GSTAMP(k) do { } while (0)
#endif

namespace {

using atmvfi::GemmDev;

constexpr float LO_UNSCALE = 1.0f / 1024.0f;
__device__ __forceinline__ int swz64(int row) { return ((row >> 2) & 1) << 1; }

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

template <int WN>
__global__ __launch_bounds__(512) void gemm_f16x3_kernel(const GemmDev a) {
    fp16_saturate_on();
    constexpr int BM = 256;
    constexpr int BN = 16 * WN;
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
    _Float16* a_hi = smem;                           // [2][BM][32]
    _Float16* a_lo = a_hi + 2 * BM * 32;
    _Float16* b_hi = a_lo + 2 * BM * 32;             // [2][BN][32]
    _Float16* b_lo = b_hi + 2 * BN * 32;
    constexpr int CF = atmvfi::gemm_const_floats(BN);
    float* cst_base = reinterpret_cast<float*>(b_lo + 2 * BN * 32);      // [2][CF] epilogue constants, by tile parity (gemm_common.h)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 15;
    const int g = lane >> 4;
    // ---- A tasks: T = tid + 512*i -> (row = T>>2, 8-channel group q = T&3); weight pieces: P = tid + 512*k ----
    const int q = tid & 3;
    const float* rbase[2];
    int iy0[2], ix0[2], adst[2];
    bool rok[2];
    // weights by LDS-DMA (k-step-major planes: one wave-instruction = 16 rows x 64 B = one contiguous KiB; no register staging,
    // no ds_write): wave-piece qq = wave + 8k -> plane = qq / WN, row group qq % WN; lane = (row in group, physical slot)
    constexpr int NWPIECE = 2 * WN;
    constexpr int NWP = (NWPIECE + 7) / 8;
    const _Float16* wsrc[NWP];
    int wdst[NWP];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (tid >> 2) + 128 * i;
        adst[i] = row * 32 + ((q ^ swz64(row)) << 3);
    }
#pragma unroll
    for (int k = 0; k < NWP; ++k) {
        const int qq = wave + 8 * k;
        const int qc = qq < NWPIECE ? qq : 0;
        wdst[k] = (qc / WN) * (2 * BN * 32) + 16 * (qc % WN) * 32;        // halves from b_hi (b_lo = b_hi + 2*BN*32)
    }
    // XCD-aware tile order.  Virtual blocks b and b+8 share an XCD (and its 4 MiB L2), so the NB column blocks that
    // re-read one 256-row activation tile are dealt to ONE XCD back to back: the tile comes from HBM once and
    // from that L2 NB-1 times.  The grid is persistent (a multiple of 8 workgroups, each walking b, b+grid, ...), so
    // a workgroup stays on the XCD its tiles were dealt to.
    auto tile_origin = [&](int vb, long long& m0, int& n0) {
        const int xcd = vb & 7;
        const int slot = vb >> 3;
        const int mgrp = slot / a.nblocks;
        const int nblk = slot - mgrp * a.nblocks;
        // each XCD owns one contiguous eighth of the row tiles: consecutive row tiles of a CONV layer read overlapping input rows
        m0 = ((long long)xcd * a.mchunk + mgrp) * BM;
        if (a.dbg & 8) m0 = ((long long)mgrp * 8 + xcd) * BM;     // A/B switch (ATMVFI_LEGACY_ORDER=1)
        n0 = nblk * BN;
    };
    auto setup_tile = [&](long long m0, int n0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (tid >> 2) + 128 * i;
            const long long m = m0 + row;
            rok[i] = m < a.M;
            const long long mm = rok[i] ? m : 0;
            if (a.mode == ATMVFI_GEMM_CONV) {
                const int hw = a.Ho * a.Wo;
                const int n = (mm >> 31) ? (int)(mm / hw) : (int)((unsigned)mm / (unsigned)hw);
                const int rem = (int)(mm - (long long)n * hw);
                const int oy = rem / a.Wo;
                const int ox = rem - oy * a.Wo;
                iy0[i] = oy * a.stride - a.pad;
                ix0[i] = ox * a.stride - a.pad;
                rbase[i] = a.in + (((long long)n * a.H + iy0[i]) * a.W + ix0[i]) * a.in_ld;
            } else {
                iy0[i] = 0;
                ix0[i] = 0;
                const long long off = (a.in_rpg > 0) ? (mm / a.in_rpg) * a.in_gstride + (mm % a.in_rpg) * (long long)a.in_ld
                                                     : mm * (long long)a.in_ld;
                rbase[i] = a.in + off;
            }
        }
#pragma unroll
        for (int k = 0; k < NWP; ++k) {
            const int qq = wave + 8 * k;
            const int qc = qq < NWPIECE ? qq : 0;
            int rg = n0 + 16 * (qc % WN);
            if (rg >= a.wrows) rg = a.wrows - 16;                          // groups past the packed rows: columns never stored
            const int row = lane >> 2;
            wsrc[k] = ((qc / WN) ? a.w_lo : a.w_hi) + (long long)(rg + row) * 32 + (((lane & 3) ^ swz64(row)) << 3);   // + kc * wrows * 32
        }
    };

    f32x4 acc[2][WN], cor[2][WN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            cor[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }

    f32x4 ra[2][2], rp[2][2];              // rp: in_prelu slopes of the same 8 channels, fetched WITH the activations
    int anv[2];

    auto load_chunk = [&](int kc) {
        const int tap = kc / a.cpt32;
        const int c0 = (kc - tap * a.cpt32) * 32;
        const int ky = tap / a.kw;
        const int kx = tap - ky * a.kw;
        const int dy = ky * a.dil, dx = kx * a.dil;
        const int c = c0 + q * 8;
        const long long toff = ((long long)dy * a.W + dx) * a.in_ld + c;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bool ok = rok[i] && (c < a.Cin);
            if (a.mode == ATMVFI_GEMM_CONV)
                ok = ok && ((unsigned)(iy0[i] + dy) < (unsigned)a.H) && ((unsigned)(ix0[i] + dx) < (unsigned)a.W);
            // unconditional loads from a clamped address + selects (a predicated load costs a branch and a vmcnt(0))
            const int nv = ok ? a.Cin - c : 0;
            const float* p = ok ? rbase[i] + toff : a.in;
            const f32x4 va = *reinterpret_cast<const f32x4*>(p);
            const f32x4 vb = *reinterpret_cast<const f32x4*>(p + (nv > 4 ? 4 : 0));
            anv[i] = nv;                       // masking / in_prelu happen at store time: no early consumer of the loads
            ra[i][0] = va;
            ra[i][1] = vb;
            if (a.in_prelu) {                  // host pads in_prelu to a multiple of 32 (uniform branch); a load issued at
                const int pc = ok ? c : 0;     // store time instead exposed an L2 round trip per chunk (deconv stages: 88 TF/s)
                rp[i][0] = *reinterpret_cast<const f32x4*>(a.in_prelu + pc);
                rp[i][1] = *reinterpret_cast<const f32x4*>(a.in_prelu + pc + 4);
            }
        }
        // this chunk's weights straight into the LDS buffer it will be read from (free: everyone passed the barrier behind its last
        // reader); waited for with the activation loads, before the barrier that publishes the chunk
        const long long koff = (long long)kc * a.wrows * 32;
#pragma unroll
        for (int k = 0; k < NWP; ++k)
            if (wave + 8 * k < NWPIECE)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[k] + koff),
                                                 (__attribute__((address_space(3))) void*)(b_hi + (kc & 1) * BN * 32 + wdst[k]), 16, 0, 0);
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f16x8 hi, lo;
            f32x4 va = ra[i][0], vb = ra[i][1];
            const int nv = anv[i];
            if (__builtin_amdgcn_ballot_w64(nv < 8)) {          // K tail, rows past M, padding taps: most waves skip the selects
                va.x = nv > 0 ? va.x : 0.f;
                va.y = nv > 1 ? va.y : 0.f;
                va.z = nv > 2 ? va.z : 0.f;
                va.w = nv > 3 ? va.w : 0.f;
                vb.x = nv > 4 ? vb.x : 0.f;
                vb.y = nv > 5 ? vb.y : 0.f;
                vb.z = nv > 6 ? vb.z : 0.f;
                vb.w = nv > 7 ? vb.w : 0.f;
            }
            if (a.in_prelu) {
                const f32x4 al = rp[i][0], bl = rp[i][1];
                va.x = va.x > 0.f ? va.x : al.x * va.x;
                va.y = va.y > 0.f ? va.y : al.y * va.y;
                va.z = va.z > 0.f ? va.z : al.z * va.z;
                va.w = va.w > 0.f ? va.w : al.w * va.w;
                vb.x = vb.x > 0.f ? vb.x : bl.x * vb.x;
                vb.y = vb.y > 0.f ? vb.y : bl.y * vb.y;
                vb.z = vb.z > 0.f ? vb.z : bl.z * vb.z;
                vb.w = vb.w > 0.f ? vb.w : bl.w * vb.w;
            }
            split8(va, vb, hi, lo);
            *reinterpret_cast<f16x8*>(a_hi + buf * BM * 32 + adst[i]) = hi;
            *reinterpret_cast<f16x8*>(a_lo + buf * BM * 32 + adst[i]) = lo;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // own weight DMA pieces of this chunk have landed (the barrier publishes them)
    };

    // Persistent tile loop.  The first chunk of the NEXT tile is fetched (registers) before the epilogue of the current
    // one, so its HBM round trip runs under the epilogue's stores, and those drain under the next k-loop.
#ifdef ATMVFI_STAMP
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
    long long m0;
    int n0;
    int vb = blockIdx.x;
    for (;; vb += gridDim.x) {                 // skip virtual blocks past the row count (grid padding to 8 XCDs)
        if (vb >= a.vblocks) return;
        tile_origin(vb, m0, n0);
        if (m0 < a.M) break;
    }
    setup_tile(m0, n0);
    int par = 0;
    atmvfi::gemm_dma_consts<BN>(a, n0, cst_base, wave, lane);
    load_chunk(0);
    GSTAMP(0);
    for (;;) {
        const float* cst = cst_base + par * CF;
        store_chunk(0);
        __syncthreads();
        GSTAMP(1);
        for (int kc = 0; kc < a.nchunks32; ++kc) {
            const int buf = kc & 1;
            if (kc + 1 < a.nchunks32) load_chunk(kc + 1);
            GSTAMP(2);
            f16x8 xh[2], xl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = 32 * wave + 16 * i + r;
                const int off = buf * BM * 32 + row * 32 + ((g ^ swz64(row)) << 3);
                xh[i] = *reinterpret_cast<const f16x8*>(a_hi + off);
                xl[i] = *reinterpret_cast<const f16x8*>(a_lo + off);
            }
            // weight fragments run two n-tiles ahead of their MFMAs (3-slot ring, static indices, order pinned): left
            // alone, hipcc folds the ring into one register and stalls on a just-issued ds_read every 6 MFMAs
            const int wbase = buf * BN * 32 + r * 32 + ((g ^ swz64(r)) << 3);      // swz64(16j + r) == swz64(r)
            f16x8 wh[3], wl[3];
            wh[0] = *reinterpret_cast<const f16x8*>(b_hi + wbase);
            wl[0] = *reinterpret_cast<const f16x8*>(b_lo + wbase);
            if (WN > 1) {
                wh[1] = *reinterpret_cast<const f16x8*>(b_hi + wbase + 16 * 32);
                wl[1] = *reinterpret_cast<const f16x8*>(b_lo + wbase + 16 * 32);
            }
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                if (j + 2 < WN) {
                    wh[(j + 2) % 3] = *reinterpret_cast<const f16x8*>(b_hi + wbase + (j + 2) * 16 * 32);
                    wl[(j + 2) % 3] = *reinterpret_cast<const f16x8*>(b_lo + wbase + (j + 2) * 16 * 32);
                }
                const f16x8 ch = wh[j % 3], cl = wl[j % 3];
                // dependent MFMAs (same accumulator) are kept 4 issues apart
                cor[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cl, xh[0], cor[0][j], 0, 0, 0);
                cor[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cl, xh[1], cor[1][j], 0, 0, 0);
                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch, xh[0], acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch, xh[1], acc[1][j], 0, 0, 0);
                cor[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch, xl[0], cor[0][j], 0, 0, 0);
                cor[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch, xl[1], cor[1][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            GSTAMP(3);
            if (kc + 1 < a.nchunks32) store_chunk(buf ^ 1);
            GSTAMP(4);
            __syncthreads();
            GSTAMP(5);
        }

        // output rows of THIS tile, then the next tile's origin, task set-up and first chunk (in flight under the stores)
        float* orow[2];
        const float* rrow[2];
        long long prow[2];
        int pc0[2];
        bool live[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long long m = m0 + 32 * wave + 16 * i + r;
            live[i] = m < a.M && atmvfi::gemm_out_row(a, m, orow[i], rrow[i], prow[i], pc0[i]);
        }
        const int n0_cur = n0;
        bool more = false;
        for (vb += gridDim.x; vb < a.vblocks; vb += gridDim.x) {
            tile_origin(vb, m0, n0);
            if (m0 < a.M) { more = true; break; }
        }
        if (more) {
            setup_tile(m0, n0);
            // next tile's constants into the other parity buffer: it was last read in the previous tile's epilogue, and every
            // wave has passed this tile's barriers since
            atmvfi::gemm_dma_consts<BN>(a, n0, cst_base + (par ^ 1) * CF, wave, lane);
            load_chunk(0);
        }
        GSTAMP(6);
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int cl = 16 * j + 4 * g;
            const atmvfi::ChanPos cp = atmvfi::gemm_chan_pos(a, n0_cur + cl);
            const f32x4 b = *reinterpret_cast<const f32x4*>(cst + cl);
            const f32x4 p = *reinterpret_cast<const f32x4*>(cst + BN + cl);
            f32x4 res[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                res[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (a.residual && live[i]) res[i] = atmvfi::gemm_load_residual4(rrow[i], cp);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (live[i]) atmvfi::gemm_finish_store4(a, orow[i], prow[i], pc0[i], cp, acc[i][j] + cor[i][j] * LO_UNSCALE, b, p, res[i]);
                acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                cor[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        par ^= 1;
        GSTAMP(7);
#ifdef ATMVFI_STAMP
        if (!more && a.stamp && lane == 0) {
            unsigned long long* o = a.stamp + ((long long)blockIdx.x * 8 + wave) * 8;
            for (int k = 0; k < 8; ++k) o[k] = tsum[k];
        }
#endif
        if (!more) return;
    }
}

template <int WN>
int launch(const GemmDev& d, int ntiles, hipStream_t s) {
    constexpr int BN = 16 * WN;
    const size_t lds = (size_t)(4 * 256 * 32 + 4 * BN * 32) * sizeof(_Float16) + 2 * atmvfi::gemm_const_floats(BN) * sizeof(float);
    auto kern = gemm_f16x3_kernel<WN>;
    const hipError_t attr_err = atmvfi::allow_dynamic_lds<gemm_f16x3_kernel<WN>>(lds);
    ATMVFI_REQUIRE(attr_err == hipSuccess, ATMVFI_ELAUNCH, "gemm_f16x3: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    GemmDev dd = d;
    dd.nblocks = (ntiles + WN - 1) / WN;
    const long long mgroups = (atmvfi::ceil_div64(d.M, 256) + 7) / 8;
    ATMVFI_REQUIRE(mgroups * 8 * dd.nblocks < (1LL << 31), ATMVFI_EINVAL, "gemm_f16x3: too many tiles");
    dd.vblocks = (int)(mgroups * 8 * dd.nblocks);
    dd.mchunk = (int)mgroups;
#if defined(ATMVFI_ABLATE) || defined(ATMVFI_STAMP)
    static const int legacy = [] { const char* e = getenv("ATMVFI_LEGACY_ORDER"); return e ? atoi(e) : 0; }();     // diagnostic builds only
    dd.dbg = legacy ? 8 : 0;
#else
    dd.dbg = 0;
#endif
#ifdef ATMVFI_STAMP
    dd.stamp = g_gemm_stamp;
#endif
    // persistent: one workgroup per CU (LDS and registers allow no more), a multiple of 8 so that b mod 8 is kept
    const int ncu = atmvfi::cu_count();
    dim3 grid((unsigned)(dd.vblocks < ncu ? dd.vblocks : ncu));
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, dd);
    return atmvfi::check_launch("gemm_f16x3");
}

}  // namespace

int atmvfi::launch_gemm_f16x3(const GemmDev& d, int ngemm, hipStream_t s) {
    const int ntiles = (ngemm + 15) / 16;
    // tile width: time ~ rounds x tile time; tile time ~ WN * (1 + cfac/WN) (MFMA work ~ WN; operand staging incl. the fp32 ->
    // fp16-pair split, redone per column block, ~ const: weight 2 measured best, deconv 788x389 -12 %); rounds = ceil(tiles / CUs)
    // of the persistent grid, which is what keeps small maps busy (M = 8 640, N = 256: 68 tiles at WN = 8, 204 at WN = 3)
#if defined(ATMVFI_ABLATE) || defined(ATMVFI_STAMP)
    static const float cfac = [] { const char* e = getenv("ATMVFI_WN_COST"); return e ? (float)atof(e) : 2.0f; }();     // diagnostic builds only
#else
    constexpr float cfac = 2.0f;
#endif
    const int ncu = atmvfi::cu_count();
    const long long mtiles = atmvfi::ceil_div64(d.M, 256);
    int best = 1;
    float best_cost = 1e30f;
    for (int wn = 1; wn <= 8; ++wn) {
        const long long tiles = mtiles * ((ntiles + wn - 1) / wn);
        const float rounds = (float)((tiles + ncu - 1) / ncu);
        const float cost = rounds * (float)wn * (1.0f + cfac / (float)wn);
        if (cost <= best_cost) { best_cost = cost; best = wn; }
    }
    if (d.force_wn > 0) best = d.force_wn;          // per-call override (atmvfi_gemm_params.tile_wn: sweeps, tests)
    switch (best) {
        case 1: return launch<1>(d, ntiles, s);
        case 2: return launch<2>(d, ntiles, s);
        case 3: return launch<3>(d, ntiles, s);
        case 4: return launch<4>(d, ntiles, s);
        case 5: return launch<5>(d, ntiles, s);
        case 6: return launch<6>(d, ntiles, s);
        case 7: return launch<7>(d, ntiles, s);
        default: return launch<8>(d, ntiles, s);
    }
}

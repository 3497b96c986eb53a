// LDS-DMA GEMM on split-plane activations, TWO WORKGROUPS PER CU (gfx950): the same contract and the same arithmetic as gemm_pp.hip
// (x = hi + lo'/1024 in fp16, three v_mfma_f32_16x16x32_f16 per product into two fp32 accumulators, k ascending, per accumulator the same
// order of products: bit-identical results), the same 64 x 64 wave tile and the same epilogue code -- but 128 x 128 tiles on 256 threads,
// two 32-KiB LDS stages, one tile per workgroup and no phase control: two workgroups share a CU, so one's prologue, fragment reads and
// epilogue (a third of a K = 384 tile in gemm_pp.hip, where all eight waves of the CU are in the epilogue together) run under the other's
// MFMAs.  Chosen per launch by launch_gemm_split (gemm_split.hip); atmvfi_gemm_params.tile_wn = -2 / -4 force its 128- / 64-column form
// (A/B against gemm_pp.hip: tools/bench_split_ab.py, tools/duo_rule.py, tools/narrow_ab.py).
#include "common.h"
#include "gemm_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

using atmvfi::GemmDev;

constexpr float LO_UNSCALE = 1.0f / 1024.0f;
constexpr int BM = 128;
// BN = 128: 4 waves as 2 x 2 of 64 x 64; BN = 64 (layers of at most 64 columns: half of a 128-column tile would be padding): 4 waves as
// 4 x 1 of 32 x 64, 24-KiB stages, three workgroups per CU
constexpr int stage_bytes(int BN) { return (2 * BM + 2 * BN) * 64; }   // [A hi 128 rows][A lo][W hi BN rows][W lo], 64 B per row
constexpr int A_LO = BM * 64, W_HI = 2 * BM * 64;
constexpr int cst_floats(int BN) { return atmvfi::gemm_const_floats(BN) + BM; }      // bias / slope of the column block + the tile's row-map entries

__device__ __forceinline__ int swz64(int row) { return ((row >> 2) & 1) << 1; }
__device__ __forceinline__ void dma16(const unsigned char* src, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void lds_write16f(unsigned addr, const f32x4& v) {
    asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
#ifdef ATMVFI_STAMP
// Diagnostic build only (`make stamp`, tools/stamp_duo.py): shader-clock sums per phase of the k-loop, per wave
static unsigned long long* g_duo_stamp = nullptr;
extern "C" void atmvfi_debug_set_duo_stamp_buffer(void* p) { g_duo_stamp = (unsigned long long*)p; }
#define DUO_SUB(k) do { if (a.stamp) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); sub[k] += t_ - sub_t; sub_t = t_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define DUO_SUB(k) do { } while (0)
#endif

template <int OFF>
__device__ __forceinline__ void lds_read16f(f32x4& d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}

template <bool CONVM, int BN>
__global__ __launch_bounds__(256, 2) void gemm_duo_kernel(const GemmDev a) {
    constexpr int STAGE = stage_bytes(BN), W_LO = W_HI + BN * 64;
    constexpr int MI = BN == 128 ? 4 : 2;           // 16-row MFMA tiles per wave along M
    constexpr int NWN = BN / 64;                    // waves along N (64 columns each)
    constexpr int WROWS = 16 * MI;
    constexpr int PIW = BN / 64;                    // 16-row DMA pieces of a W plane per wave (A: always 2)
    fp16_saturate_on();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15;
    const int g = lane >> 4;
    const int wm = wave / NWN, wn = wave % NWN;

    // XCD-aware tile order of gemm_pp.hip (virtual block v: xcd = v & 7, slot = v >> 3, row tile = (slot / nblocks) * 8 + xcd, column
    // block = slot % nblocks: the column blocks of one row tile meet in one L2), one tile per workgroup
    const int M = (int)a.M;
    int m0, n0;
    {
        const int v = blockIdx.x;
        const int xcd = v & 7, slot = v >> 3;
        const int mgrp = slot / a.nblocks;
        m0 = (mgrp * 8 + xcd) * BM;
        n0 = (slot - mgrp * a.nblocks) * BN;
    }
    if (m0 >= M) return;
    float* cst_w = reinterpret_cast<float*>(smem + 2 * STAGE);
    const float* cst = cst_w;

    // ---- DMA pieces of this wave: A rows (i * 4 + wave) * 16 + lane / 4 (i = 0, 1) of both planes, W rows the same.  LDS image of a
    // piece: wave-uniform base + lane * 16 B; the slot swizzle goes on the SOURCE address.
    const unsigned ls16 = (unsigned)(((lane & 3) ^ swz64(lane >> 2)) << 4);
    unsigned aoff[2], wsoff[2];
    int crow[2], cyx[2];
    int c_ky = 0, c_kx = 0, c_chunk = 0;
    const int zero_row = a.in_N * a.H * a.W;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int m = m0 + (i * 4 + wave) * 16 + (lane >> 2);
        if (m >= M) m = M - 1;
        if constexpr (CONVM) {
            const int hw = a.Ho * a.Wo, wo = a.Wo;
            const int n = (int)((unsigned)m / (unsigned)hw);
            const int rem = m - n * hw;
            const int oy = rem / wo, ox = rem - oy * wo;
            const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
            cyx[i] = (iy0 << 16) | (ix0 & 0xffff);
            crow[i] = (n * a.H + iy0) * a.W + ix0;
        } else {
            aoff[i] = (unsigned)m * 64u + ls16;
        }
        if (i < PIW) {
            int n = n0 + (i * 4 + wave) * 16 + (lane >> 2);
            if (n >= a.wrows) n = a.wrows - 1;
            wsoff[i] = (unsigned)n * 64u + ls16;
        }
    }
    const unsigned char* pa_hi = reinterpret_cast<const unsigned char*>(a.a_hi);
    const unsigned char* pa_lo = reinterpret_cast<const unsigned char*>(a.a_lo);
    const unsigned char* pw_hi = reinterpret_cast<const unsigned char*>(a.w_hi);
    const unsigned char* pw_lo = reinterpret_cast<const unsigned char*>(a.w_lo);
    const long long a_step = (long long)a.in_ld * 64, w_step = (long long)a.wrows * 64;
    int nk = a.nchunks32;
    if (a.ksplit > 1) {
        // this workgroup's range of k-steps: the first (nk mod ksplit) ranges take one more
        const int sp = blockIdx.y;
        const int per = nk / a.ksplit, rem = nk - per * a.ksplit;
        const int u0 = sp * per + (sp < rem ? sp : rem);
        nk = per + (sp < rem ? 1 : 0);
        pw_hi += (long long)u0 * w_step;
        pw_lo += (long long)u0 * w_step;
        if constexpr (CONVM) {
            const int tap = u0 / a.cpt32;
            c_chunk = u0 - tap * a.cpt32;
            c_ky = tap / a.kw;
            c_kx = tap - c_ky * a.kw;
        } else {
            pa_hi += (long long)u0 * a_step;
            pa_lo += (long long)u0 * a_step;
        }
    }
    auto issue_stage = [&](int buf) {                               // 4 + 2 PIW pieces per wave
        unsigned char* dst = smem + buf * STAGE + wave * 1024;
        if constexpr (CONVM) {
            const bool second = c_chunk >= a.split_chunks;
            const unsigned char* bh = second ? reinterpret_cast<const unsigned char*>(a.a_hi2) + (long long)(c_chunk - a.split_chunks) * a.in_ld2 * 64
                                             : pa_hi + (long long)c_chunk * a_step;
            const unsigned char* bl = second ? reinterpret_cast<const unsigned char*>(a.a_lo2) + (long long)(c_chunk - a.split_chunks) * a.in_ld2 * 64
                                             : pa_lo + (long long)c_chunk * a_step;
            const int dyo = c_ky * a.dil, dxo = c_kx * a.dil;
            unsigned off[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int iy = (cyx[i] >> 16) + dyo, ix = (int)(short)cyx[i] + dxo;
                const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const int row = ok ? crow[i] + dyo * a.W + dxo : zero_row;
                off[i] = (unsigned)row * 64u + ls16;
            }
            dma16(bh + off[0], dst);
            dma16(bh + off[1], dst + 4 * 1024);
            dma16(bl + off[0], dst + A_LO);
            dma16(bl + off[1], dst + A_LO + 4 * 1024);
            if (++c_chunk == a.cpt32) {
                c_chunk = 0;
                if (++c_kx == a.kw) { c_kx = 0; ++c_ky; }
            }
        } else {
            dma16(pa_hi + aoff[0], dst);
            dma16(pa_hi + aoff[1], dst + 4 * 1024);
            dma16(pa_lo + aoff[0], dst + A_LO);
            dma16(pa_lo + aoff[1], dst + A_LO + 4 * 1024);
            pa_hi += a_step;
            pa_lo += a_step;
        }
        dma16(pw_hi + wsoff[0], dst + W_HI);
        if constexpr (PIW == 2) dma16(pw_hi + wsoff[1], dst + W_HI + 4 * 1024);
        dma16(pw_lo + wsoff[0], dst + W_LO);
        if constexpr (PIW == 2) dma16(pw_lo + wsoff[1], dst + W_LO + 4 * 1024);
        pw_hi += w_step;
        pw_lo += w_step;
    };
    // per-tile constants by LDS-DMA, one piece per wave (4 waves: bias and slope halves; the row map's 128 entries by waves 0-1 again)
    {
        atmvfi::gemm_dma_consts<BN>(a, n0, cst_w, wave, lane);
        if (a.out_row_map && a.mode == ATMVFI_GEMM_LINEAR) {
            int m = m0 + (wave & 1) * 64 + lane;
            if (m >= M) m = M - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.out_row_map + m),
                                             (__attribute__((address_space(3))) void*)(cst_w + atmvfi::gemm_const_floats(BN) + (wave & 1) * 64), 4, 0, 0);
        } else {
            atmvfi::gemm_dma_consts<BN>(a, n0, cst_w, wave, lane);
        }
    }

    f32x4 acc[MI][4], cor[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            cor[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#ifdef ATMVFI_STAMP
    unsigned long long sub[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sub_t = 0, t_begin = 0;
    if (a.stamp) { t_begin = sub_t = __builtin_amdgcn_s_memtime(); }
#endif
    issue_stage(0);
    if (nk > 1) issue_stage(1);
    DUO_SUB(0);
    f16x8 xh[MI], xl[MI], wh[4], wl[4];
    const unsigned xfrag = lds_offset(smem) + (unsigned)((WROWS * wm + r) * 64 + ((g ^ swz64(r)) << 4));
    const unsigned wfrag = lds_offset(smem) + (unsigned)(W_HI + (64 * wn + r) * 64 + ((g ^ swz64(r)) << 4));
    for (int u = 0; u < nk; ++u) {
        // this wave's pieces of stage u have landed (8 newer ones, stage u + 1, may stay in flight), then everybody's
        if (u + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + 2 * PIW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        DUO_SUB(1);
        const unsigned off = (unsigned)((u & 1) * STAGE);
        static_for<0, MI>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            lds_read16<i * 1024>(xh[i], xfrag + off);
            lds_read16<i * 1024 + A_LO>(xl[i], xfrag + off);
        });
        static_for<0, 4>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            lds_read16<j * 1024>(wh[j], wfrag + off);
            lds_read16<j * 1024 + BN * 64>(wl[j], wfrag + off);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xh[0]), "+v"(xh[1]), "+v"(xl[0]), "+v"(xl[1]));
        if constexpr (MI == 4) asm volatile("" : "+v"(xh[2]), "+v"(xh[3]), "+v"(xl[2]), "+v"(xl[3]));
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(wh[j]), "+v"(wl[j]));
        DUO_SUB(2);
        __builtin_amdgcn_s_barrier();                   // every wave has its fragments: the buffer may be refilled
        DUO_SUB(3);
        if (u + 2 < nk) issue_stage(u & 1);
        DUO_SUB(4);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                cor[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[j], xh[i], cor[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xh[i], acc[i][j], 0, 0, 0);
            }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) cor[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xl[i], cor[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        DUO_SUB(5);
    }
    {
        __builtin_amdgcn_s_barrier();                       // every wave is past its last fragment read (the transposition reuses stage 0)
        // ---- epilogue.  After the k-loop lane (r, g) holds, of its 16 MFMA tiles (i, j), GEMM row 64 wm + 16 i + r and columns
        // 64 wn + 16 j + 4 g .. + 3: a 16-byte store per lane in that layout touches 16 rows x 64 B per instruction, and the store
        // path then takes 52 cycles per instruction where 256 contiguous bytes per 16 lanes take 15 (tools/probes/store_probe.hip:
        // 19.6 against 67.5 B/clk/CU; 8 rows x 128 B is no better than 16 x 64 B) -- 8 k of a 42 k-cycle K = 384 tile.  So every
        // 16-row slab i is TRANSPOSED THROUGH LDS, in place: four ds_write_b128 (lane (r, g): row r, 16-byte slot 4 j + g) and four
        // ds_read_b128 (lane (r, g): row 4 q + g, slot r) leave lane (r, g) with GEMM rows 64 wm + 16 i + 4 q + g, q = 0..3, and
        // columns 64 wn + 4 r .. + 3 -- the 16 lanes of a row cover 256 contiguous bytes, residual loads included.  The 4 KiB a wave
        // needs are ITS OWN four A pieces of the stage buffer the last k-step was read from: nobody reads that buffer any more (this
        // group's epilogue starts a barrier after the other group's last read), and the only DMA that can land there before this wave
        // is done is the wave's own next issue.  Slot XOR-swizzle (physical slot = slot ^ row): conflict-free for the 8-lane
        // groups of ds_write_b128 (banks mod 32) and the 16-lane groups of ds_read_b128 (MI355X_MICROARCH.md, LDS).
        // Per-tile constants: bias / slope of the lane's four columns (two LDS reads per tile instead of two per vector); row-map
        // entries from LDS; (LINEAR with a residual whose width is a multiple of 4: every case of the network) all sixteen
        // residual vectors in one batch of unconditional loads before the first store (one wait per tile; vmcnt counts stores on
        // gfx9, so a wait per row would drain the previous row's stores).
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = acc[i][j] + cor[i][j] * LO_UNSCALE;
                asm volatile("" : "+v"(acc[i][j]));
            }
        {
            // 4 KiB per wave in stage buffer 0 (every wave is past its last fragment read: barrier above): slab rows 256 B apart
            const unsigned tb = lds_offset(smem) + (unsigned)(wave * 4096);
            const unsigned wbase = tb + (unsigned)(r * 256);
            const unsigned rbase = tb + (unsigned)(g * 256);
            unsigned wad[4], rad[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                wad[k] = wbase + (unsigned)((((4 * k + g) ^ r) & 15) << 4);               // row r, slot (4 j + g) ^ r
                rad[k] = rbase + (unsigned)(k * 1024) + (unsigned)(((r ^ (4 * k + g)) & 15) << 4);   // row 4 q + g, slot r ^ (4 q + g)
            }
            static_for<0, MI>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                static_for<0, 4>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    lds_write16f(wad[j], acc[i][j]);
                });
                static_for<0, 4>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    lds_read16f<0>(acc[i][q], rad[q]);
                });
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(acc[i][q]));
        }
        if (a.ksplit > 1) {
            // split-K: the raw sums of this K range as plain rows of the partial buffer; the epilogue runs in the reduce kernel
            float* pb = a.part + blockIdx.y * a.part_stride + (n0 + 64 * wn + 4 * r);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = m0 + WROWS * wm + g + 16 * i + 4 * q;
                    if (m < M) *reinterpret_cast<f32x4*>(pb + (long long)m * a.part_ld) = acc[i][q];
                }
            return;
        }
        const bool mapped = a.out_row_map && a.mode == ATMVFI_GEMM_LINEAR;
        const int cl = 64 * wn + 4 * r;                          // the lane's four columns inside the column block
        const int nb = n0 + cl;
        const atmvfi::ChanPos cp = atmvfi::gemm_chan_pos(a, nb);
        const f32x4 bvec = *reinterpret_cast<const f32x4*>(cst + cl);
        const f32x4 pvec = *reinterpret_cast<const f32x4*>(cst + BN + cl);
        const int mrow = m0 + WROWS * wm + g;                       // row of (i, q) = (0, 0); (i, q) adds 16 i + 4 q
        int ro[MI][4];              // output row: the row map's entry, or (unmapped) the row itself; < 0: nothing to store
        if (mapped) {                 // (the test outside the loops: inside, hipcc makes it a scalar branch per row)
            const int* mp = reinterpret_cast<const int*>(cst + atmvfi::gemm_const_floats(BN)) + WROWS * wm + g;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) ro[i][q] = (mrow + 16 * i + 4 * q) < M ? mp[16 * i + 4 * q] : -1;
        } else {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = mrow + 16 * i + 4 * q;
                    ro[i][q] = m < M ? m : -1;
                }
        }
        const bool vec_res = !CONVM && a.residual && (a.Cout & 3) == 0 && a.mode != ATMVFI_GEMM_DECONV;
        f32x4 res[MI][4];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) res[i][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (vec_res && a.fit32 && m0 + BM <= M) {
            // (all rows live: 32-bit offsets from a scalar base, stepped by 4 rows)
            const unsigned char* rb = reinterpret_cast<const unsigned char*>(a.residual);
            unsigned voff = nb < a.Cout ? ((unsigned)mrow * (unsigned)a.res_ld + (unsigned)nb) * 4u : 0u;
            const unsigned step = nb < a.Cout ? (unsigned)a.res_ld * 16u : 0u;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    res[i][q] = *reinterpret_cast<const f32x4*>(rb + voff);
                    voff += step;
                }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(res[i][q]));
        } else if (vec_res) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = mrow + 16 * i + 4 * q;
                    const float* p = (ro[i][q] >= 0 && nb < a.Cout) ? a.residual + m * (long long)a.res_ld + nb : a.residual;
                    res[i][q] = *reinterpret_cast<const f32x4*>(p);
                }
            // every vector is "used" here, dead rows' too: a load left pending on some path makes hipcc guard the next tile's first
            // write to its register with s_waitcnt vmcnt(0), which would also wait for this tile's stores
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(res[i][q]));
        }
        // DECONV: the lane's rows are 4 apart: (image, y, x) of the first one by division, the others by stepping
        int dn = 0, dy = 0, dx = 0;
        if (a.mode == ATMVFI_GEMM_DECONV) {
            const int hw = a.H * a.W;
            dn = (int)((unsigned)mrow / (unsigned)hw);
            const int rem = mrow - dn * hw;
            dy = rem / a.W;
            dx = rem - dy * a.W;
        }
        // (two copies of the store loop under a uniform branch: with the ragged-width residual loads as a conditional inside one
        // loop, hipcc puts their s_waitcnt vmcnt(0) into the shared block, i.e. in front of every store of the common case too)
        auto store_rows = [&](auto ragged_tag) {
            constexpr bool RAGGED = decltype(ragged_tag)::value;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = mrow + 16 * i + 4 * q;
                    float* orow = nullptr;
                    long long prow = 0;
                    int pc0 = a.out_c0;
                    if (a.mode == ATMVFI_GEMM_DECONV) {
                        prow = ((long long)dn * a.Ho + 2 * dy) * a.Wo + 2 * dx;
                        orow = a.out + prow * a.out_ld;
                        dx += 4;                                  // next row of this lane
                        while (dx >= a.W) { dx -= a.W; if (++dy == a.H) { dy = 0; ++dn; } }
                    } else {
                        const int rr = ro[i][q] < 0 ? 0 : ro[i][q];
                        prow = rr;
                        long long off = (long long)rr * a.out_ld;
                        if (a.out_rpg > 0) {
                            const int gi = (int)((unsigned)rr / (unsigned)a.out_rpg);
                            prow = rr - gi * a.out_rpg;
                            off = gi * a.out_gstride + prow * (long long)a.out_ld;
                            pc0 += gi * a.out_gc;
                        }
                        orow = a.out + off;
                    }
                    if (ro[i][q] >= 0) {
                        f32x4 rv = res[i][q];
                        if constexpr (RAGGED) rv = atmvfi::gemm_load_residual4(a.residual + m * (long long)a.res_ld, cp);
                        atmvfi::gemm_finish_store4(a, orow, prow, pc0, cp, acc[i][q], bvec, pvec, rv);
                    }
                }
        };
        // Fast variants for the network's three output shapes, chosen once per tile: inside them nothing is decided per row (the
        // generic loop above spends ~10 scalar / exec branches per row on mode, groups, sinks and ragged widths: 11 k cycles per
        // tile for 16 rows per lane).  All have full 4-channel vectors (Cout % 4 == 0, or DECONV whose position blocks are padded
        // to 4) and do the arithmetic of gemm_finish_store4 in its order: + bias, PReLU (slope 1 when absent), + residual.
        // (the fast loops exist twice, with and without a PReLU: the select costs 14 of a row's ~30 VALU instructions, and the
        // epilogue of a group is VALU-bound -- one wave per SIMD beside the other group's MFMAs)
        auto fast_rows = [&](auto prelu_tag) {
            constexpr bool PRELU = decltype(prelu_tag)::value;
            auto finish = [&](int i, int q, bool with_res = true) -> f32x4 {
                f32x4 v = acc[i][q] + bvec;
                if constexpr (PRELU) {
                    v.x = v.x > 0.f ? v.x : pvec.x * v.x;
                    v.y = v.y > 0.f ? v.y : pvec.y * v.y;
                    v.z = v.z > 0.f ? v.z : pvec.z * v.z;
                    v.w = v.w > 0.f ? v.w : pvec.w * v.w;
                }
                if constexpr (CONVM) return v;           // convolutions have no residual operand
                else return with_res ? v + res[i][q] : v;
            };
            if (a.mode != ATMVFI_GEMM_DECONV && !a.out_hi && a.fit32 && m0 + BM <= M) {
                // fp32 rows, every row of the tile live, no map: one 32-bit byte offset per lane, stepped by 4 rows -- a row costs
                // its arithmetic (bias, residual) and one add; stores with a scalar base
                if (nb < a.Cout) {
                    unsigned char* ob = reinterpret_cast<unsigned char*>(a.out);
                    unsigned voff = ((unsigned)mrow * (unsigned)a.out_ld + (unsigned)nb) * 4u;
                    const unsigned step = (unsigned)a.out_ld * 16u;
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            *reinterpret_cast<f32x4*>(ob + voff) = finish(i, q);
                            voff += step;
                        }
                }
            } else if (a.mode != ATMVFI_GEMM_DECONV && !a.out_hi) {
                // fp32 rows (qkv, fc1, proj with its row map and residual, fc2, fusion projections; convolutions)
                float* obase = a.out + nb;
                const bool col_ok = nb < a.Cout;
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = finish(i, q);
                        const unsigned long long off = (unsigned long long)(unsigned)ro[i][q] * (unsigned)a.out_ld;
                        if (ro[i][q] >= 0 && col_ok) *reinterpret_cast<f32x4*>(obase + off) = v;
                    }
            } else if (a.mode != ATMVFI_GEMM_DECONV) {
                // plane sink, with or without fp32 rows (optionally a grouped [G, R, C] view): the last fc2 of a motion branch,
                // strided / 1x1 convolutions between plane maps
                const unsigned rpg = a.out_rpg > 0 ? (unsigned)a.out_rpg : 0x7fffffffu;
                const RowSink sink{nullptr, 0, a.out_hi, a.out_lo, a.out_plane_rows};
                const bool col_ok = nb < a.Cout;
                // ungrouped sink whose planes fit 32-bit byte offsets (every strided / 1x1 conv between plane maps): the lane's
                // channel group fixes a pointer into each plane once per tile, a row costs a shift (the grouped form divides
                // every row by the group size: ~25 VALU instructions of a VALU-bound epilogue)
                auto rows32 = [&](auto f32_tag) {
                    constexpr bool F32 = decltype(f32_tag)::value;
                    const int c = a.out_c0 + nb;
                    const long long cpart = ((long long)(c >> 5) * a.out_plane_rows) * 32 + (c & 31);
                    unsigned char* ph = reinterpret_cast<unsigned char*>(a.out_hi + cpart);
                    unsigned char* pl = reinterpret_cast<unsigned char*>(a.out_lo + cpart);
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 v = finish(i, q);
                            const unsigned rr = ro[i][q] < 0 ? 0u : (unsigned)ro[i][q];
                            if (ro[i][q] >= 0 && col_ok) {
                                if constexpr (F32) *reinterpret_cast<f32x4*>(a.out + (unsigned long long)rr * (unsigned)a.out_ld + nb) = v;
                                f16x2 h0, l0, h1, l1;
                                split_pair((f32x2){v.x, v.y}, h0, l0);
                                split_pair((f32x2){v.z, v.w}, h1, l1);
                                *reinterpret_cast<f16x4*>(ph + (rr << 6)) = (f16x4){h0.x, h0.y, h1.x, h1.y};
                                *reinterpret_cast<f16x4*>(pl + (rr << 6)) = (f16x4){l0.x, l0.y, l1.x, l1.y};
                            }
                        }
                };
                auto rows = [&](auto f32_tag) {
                    constexpr bool F32 = decltype(f32_tag)::value;
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 v = finish(i, q);
                            const unsigned rr = ro[i][q] < 0 ? 0u : (unsigned)ro[i][q];
                            const unsigned gi = rr / rpg;
                            const long long prow = rr - gi * rpg;
                            if (ro[i][q] >= 0 && col_ok) {
                                if constexpr (F32) *reinterpret_cast<f32x4*>(a.out + gi * a.out_gstride + prow * (long long)a.out_ld + nb) = v;
                                sink_store4(sink, prow, a.out_c0 + (int)gi * a.out_gc + nb, v);
                            }
                        }
                };
                if (a.out_rpg == 0 && a.pfit32) {
                    if (a.out) rows32(std::true_type{});
                    else rows32(std::false_type{});
                } else if (a.out) rows(std::true_type{});
                else rows(std::false_type{});
            } else {
                // ConvTranspose2d 2x2 / stride 2 into a plane sink (decoder and U-Net stages): column -> (position, channel) is a
                // lane constant; the lane's rows are 4 input pixels apart: (image, y, x) of the first by division, the others by
                // stepping
                const RowSink sink{nullptr, 0, a.out_hi, a.out_lo, a.out_plane_rows};
                const int hw = a.H * a.W;
                int dn = (int)((unsigned)mrow / (unsigned)hw);
                const int rem = mrow - dn * hw;
                int dy = rem / a.W;
                int dx = rem - dy * a.W;
                const int qoff = (cp.q >> 1) * a.Wo + (cp.q & 1);
                const int pc = a.out_c0 + cp.co;
                if (a.pfit32) {
                    // 32-bit row offsets from the lane's chunk pointers.  Output row index of (image dn, input row dy, input
                    // column 0) is (dn Ho + 2 dy) Wo, and because Ho = 2 H it simply grows by 2 Wo whenever the stepped input
                    // column wraps -- across images too: no (dn, dy) bookkeeping, no 64-bit products per row.
                    const long long cpart = ((long long)(pc >> 5) * a.out_plane_rows) * 32 + (pc & 31);
                    unsigned char* ph = reinterpret_cast<unsigned char*>(a.out_hi + cpart);
                    unsigned char* pl = reinterpret_cast<unsigned char*>(a.out_lo + cpart);
                    unsigned rowbase = ((unsigned)dn * (unsigned)a.Ho + 2u * (unsigned)dy) * (unsigned)a.Wo + (unsigned)qoff;
                    const unsigned rstep = 2u * (unsigned)a.Wo;
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            // (no residual on this path; channels past Cout inside the group of 4 are already zero: their weight
                            // rows and their bias are, and PReLU keeps a zero)
                            const f32x4 v = finish(i, q, false);
                            const unsigned boff = (rowbase + 2u * (unsigned)dx) << 6;
                            if (ro[i][q] >= 0 && cp.nvalid > 0) {
                                f16x2 h0, l0, h1, l1;
                                split_pair((f32x2){v.x, v.y}, h0, l0);
                                split_pair((f32x2){v.z, v.w}, h1, l1);
                                *reinterpret_cast<f16x4*>(ph + boff) = (f16x4){h0.x, h0.y, h1.x, h1.y};
                                *reinterpret_cast<f16x4*>(pl + boff) = (f16x4){l0.x, l0.y, l1.x, l1.y};
                            }
                            dx += 4;                                  // next row of this lane (W >= 4: one wrap at most)
                            const bool wrap = dx >= a.W;
                            dx = wrap ? dx - a.W : dx;
                            rowbase += wrap ? rstep : 0u;
                        }
                } else
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v = finish(i, q);
                        v.y = cp.nvalid > 1 ? v.y : 0.f;            // channels past Cout inside the group of 4: the planes' pad channels
                        v.z = cp.nvalid > 2 ? v.z : 0.f;
                        v.w = cp.nvalid > 3 ? v.w : 0.f;
                        const long long prow = ((long long)dn * a.Ho + 2 * dy) * a.Wo + 2 * dx + qoff;
                        if (ro[i][q] >= 0 && cp.nvalid > 0) sink_store4(sink, prow, pc, v);
                        dx += 4;                                      // next row of this lane (W >= 4: one wrap at most)
                        const bool wrap = dx >= a.W;
                        dx = wrap ? dx - a.W : dx;
                        dy = wrap ? dy + 1 : dy;
                        const bool wrap2 = dy >= a.H;
                        dy = wrap2 ? 0 : dy;
                        dn = wrap2 ? dn + 1 : dn;
                    }
            }
        };
        const bool c4 = (a.Cout & 3) == 0;
        const bool fast = (a.mode != ATMVFI_GEMM_DECONV && c4 && (a.out_hi || (a.out && a.out_rpg == 0))) ||
                          (a.mode == ATMVFI_GEMM_DECONV && !a.out && a.out_hi && a.W >= 4 && !a.residual);
        if (fast) {
            if (a.prelu) fast_rows(std::true_type{});
            else fast_rows(std::false_type{});
        } else if (a.residual && !vec_res) {
            store_rows(std::true_type{});
        } else {
            store_rows(std::false_type{});
        }
    }
#ifdef ATMVFI_STAMP
    DUO_SUB(6);
    if (a.stamp && lane == 0) {
        unsigned long long* o = a.stamp + ((long long)blockIdx.x * 4 + wave) * 8;
        for (int k = 0; k < 7; ++k) o[k] = sub[k];
        o[7] = sub_t - t_begin;
    }
#endif
}

// Split-K, second half: the epilogue of the kernels above (gemm_common.h: output row of a GEMM row incl. the window-reverse map, row
// groups and the deconv's pixel shuffle; + bias, PReLU, + residual; fp32 rows and / or plane sink) on the sum of the partial buffers,
// added in split order: run-to-run deterministic.  One thread per (GEMM row, 4 columns).
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const GemmDev a, int ngemm4) {
    fp16_saturate_on();
    const long long total = a.M * (long long)ngemm4;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long m = idx / ngemm4;
        const int nb = (int)(idx - m * ngemm4) * 4;
        const atmvfi::ChanPos cp = atmvfi::gemm_chan_pos(a, nb);
        if (cp.nvalid <= 0) continue;
        // bias / slope of the four columns (selects, no indexed local arrays: hipcc turns those into scratch)
        auto vec4 = [&](const float* t, float dflt) -> f32x4 {
            if (!t) return (f32x4){dflt, dflt, dflt, dflt};
            if (cp.nvalid >= 4) return *reinterpret_cast<const f32x4*>(t + cp.co);
            f32x4 o = (f32x4){t[cp.co], dflt, dflt, dflt};
            if (cp.nvalid > 1) o.y = t[cp.co + 1];
            if (cp.nvalid > 2) o.z = t[cp.co + 2];
            return o;
        };
        const float* p = a.part + m * a.part_ld + nb;
        f32x4 v = *reinterpret_cast<const f32x4*>(p);
        for (int sp = 1; sp < a.ksplit; ++sp) v += *reinterpret_cast<const f32x4*>(p + sp * a.part_stride);
        float* orow;
        const float* rrow;
        long long prow;
        int pc0;
        if (!atmvfi::gemm_out_row(a, m, orow, rrow, prow, pc0)) continue;
        const f32x4 res = rrow ? atmvfi::gemm_load_residual4(rrow, cp) : (f32x4){0.f, 0.f, 0.f, 0.f};
        atmvfi::gemm_finish_store4(a, orow, prow, pc0, cp, v, vec4(a.bias, 0.f), vec4(a.prelu, 1.f), res);
    }
}

}  // namespace

template <bool CONVM, int BN>
static int launch_duo(const GemmDev& d, int ngemm, hipStream_t s) {
    const size_t lds = (size_t)2 * stage_bytes(BN) + cst_floats(BN) * sizeof(float);
    const hipError_t attr_err = atmvfi::allow_dynamic_lds<gemm_duo_kernel<CONVM, BN>>(lds);
    ATMVFI_REQUIRE(attr_err == hipSuccess, ATMVFI_ELAUNCH, "gemm_duo: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    ATMVFI_REQUIRE((long long)d.in_ld * 64 < (1ll << 32) && (long long)d.wrows * 64 < (1ll << 32) &&
                       (!d.a_hi2 || (long long)d.in_ld2 * 64 < (1ll << 32)) && d.M < (1ll << 26), ATMVFI_EINVAL,
                   "gemm_duo: plane rows x 64 bytes must fit 32 bits (got %d rows)", d.in_ld);
    if (CONVM) ATMVFI_REQUIRE(d.H < 32768 && d.W < 32768, ATMVFI_EINVAL, "gemm_duo: CONV mode packs (y, x) into 16 bits each");
    GemmDev dd = d;
    dd.dbg = 0;
    dd.pfit32 = d.out_hi && (long long)d.out_plane_rows * 64 < (1ll << 32);
    dd.fit32 = d.out && !d.out_row_map && d.out_rpg == 0 && d.mode != ATMVFI_GEMM_DECONV && (d.M + 1) * (long long)d.out_ld * 4 < (1ll << 32) &&
               (!d.residual || (d.M + 1) * (long long)d.res_ld * 4 < (1ll << 32));
#ifdef ATMVFI_STAMP
    dd.stamp = g_duo_stamp;
#endif
    dd.nblocks = (ngemm + BN - 1) / BN;
    const long long mgroups = (atmvfi::ceil_div64(d.M, BM) + 7) / 8;
    ATMVFI_REQUIRE(mgroups * 8 * dd.nblocks < (1LL << 31), ATMVFI_EINVAL, "gemm_duo: grid too large");
    dd.vblocks = (int)(mgroups * 8 * dd.nblocks);
    if (dd.ksplit > 1) {
        hipLaunchKernelGGL((gemm_duo_kernel<CONVM, BN>), dim3((unsigned)dd.vblocks, (unsigned)dd.ksplit), dim3(256), lds, s, dd);
        const int rc = atmvfi::check_launch("gemm_duo (split-K)");
        if (rc != ATMVFI_OK) return rc;
        const int ngemm4 = (ngemm + 3) / 4;
        const long long groups = d.M * ngemm4;
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)std::min<long long>((groups + 255) / 256, 8192)), dim3(256), 0, s, dd, ngemm4);
        return atmvfi::check_launch("gemm_duo (split-K reduce)");
    }
    hipLaunchKernelGGL((gemm_duo_kernel<CONVM, BN>), dim3((unsigned)dd.vblocks), dim3(256), lds, s, dd);
    return atmvfi::check_launch("gemm_duo");
}

int atmvfi::launch_gemm_duo(const GemmDev& d, int ngemm, hipStream_t s) {
    // at most 64 GEMM columns: 64-column tiles (tile_wn = -2 forces the 128-column form for the A/B, -4 the 64-column one)
    if ((ngemm <= 64 && d.force_wn != -2) || d.force_wn == -4)
        return d.mode == ATMVFI_GEMM_CONV ? launch_duo<true, 64>(d, ngemm, s) : launch_duo<false, 64>(d, ngemm, s);
    return d.mode == ATMVFI_GEMM_CONV ? launch_duo<true, 128>(d, ngemm, s) : launch_duo<false, 128>(d, ngemm, s);
}

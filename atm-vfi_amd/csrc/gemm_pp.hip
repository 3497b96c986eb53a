// LDS-DMA GEMM on split-plane activations, "ping-pong" schedule (gfx950): nn.Linear rows and ConvTranspose2d 2x2 / stride 2
// (attention.py:138-141, 349-351, 93-96; network_base.py:27-32, 79-84) whose input the producing layer left as split planes.
//
// Same arithmetic as gemm_split.hip / gemm_f16x3.hip (x = hi + lo'/1024 in fp16, three v_mfma_f32_16x16x32_f16 per product into two
// fp32 accumulators, k ascending, per accumulator the same order of products: bit-identical results), same operand layouts, same
// 256 x 128 tile on 512 threads (8 waves as 4 x 2, 64 x 64 each), same 48 KiB LDS stage x 3 filled by global_load_lds_dwordx4.
// What changes is WHEN things happen (the schedule conv3x3_planes.hip introduced for the 3x3 kernel):
//
//   * the eight waves form two groups (waves 0-3 / 4-7 = the two waves of each SIMD) that run ONE PHASE APART: while a group issues
//     the 48 MFMAs of k-step u from registers (s_setprio 1), its SIMD partners read the 16 fragments of their k-step from LDS
//     (ds_read_b128, one per-lane base + immediates) and issue their six DMA pieces of the stage two k-steps ahead; a raw
//     s_barrier swaps the roles.  gemm_split.hip interleaves fragment reads, DMA issue and MFMAs inside every wave and both waves
//     of a SIMD meet the same barrier in the same state: 2 400-2 700 cycles per k-step for 1 536 cycles of matrix work
//     (tools/stamp_split.py);
//   * waits: a wave's pieces of stage u+1 went out two read phases ago and six newer pieces (stage u+2) are allowed to stay in
//     flight (vmcnt(6)); the FIRST group waits at the end of its MFMA phase, the second at the end of its read phase -- the same
//     barrier -- so that stage u+1 is complete, for every wave, one barrier before its first reader (the first group) starts, and
//     both groups' pieces had 1.5-2 k-steps to land.  A stage's buffer is refilled one barrier after its last reader (the second
//     group) has drained its reads (lgkmcnt(0) before the barrier);
//   * the k-loop body has no data-dependent branch; the last two k-steps are separate copies without DMA issue, so nothing is in
//     flight when the epilogue starts.
#include "common.h"
#include "gemm_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#ifdef ATMVFI_STAMP
static unsigned long long* g_pp_stamp = nullptr;
extern "C" void atmvfi_debug_set_pp_stamp_buffer(void* p) { g_pp_stamp = (unsigned long long*)p; }
#define PP_STAMP(i) do { if (a.stamp) { tstamp[i] = __builtin_amdgcn_s_memtime(); rstamp[i] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define PP_STAMP(i) do { } while (0)
#endif

namespace {

using atmvfi::GemmDev;

constexpr float LO_UNSCALE = 1.0f / 1024.0f;
constexpr int BM = 256, BN = 128;
constexpr int STAGE = (2 * BM + 2 * BN) * 64;          // bytes: [A hi 256 rows][A lo][W hi 128 rows][W lo], 64 B per row
constexpr int A_LO = BM * 64, W_HI = 2 * BM * 64, W_LO = W_HI + BN * 64;
constexpr int PIECES = 6;                              // 16-byte DMA pieces per thread and stage: 4 of A, 2 of W
constexpr int CST_FLOATS = atmvfi::gemm_const_floats(BN) + BM;      // bias / slope of the column block + the tile's row-map entries

__device__ __forceinline__ int swz64(int row) { return ((row >> 2) & 1) << 1; }
__device__ __forceinline__ void dma16(const unsigned char* src, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(const GemmDev a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int r = lane & 15;
    const int g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile order: blocks b and b+8 share an XCD, so the column blocks of one row tile go to one L2
    const int xcd = blockIdx.x & 7;
    const int slot = blockIdx.x >> 3;
    const int mgrp = slot / a.nblocks;
    const int nblk = slot - mgrp * a.nblocks;
    const long long m0 = ((long long)mgrp * 8 + xcd) * BM;
    if (m0 >= a.M) return;
    const int n0 = nblk * BN;
    float* cst = reinterpret_cast<float*>(smem + 3 * STAGE);

    // ---- DMA pieces of this wave.  A: rows (i * 8 + wave) * 16 + lane / 4 (i = 0, 1) of the hi and of the lo plane; W: rows
    // wave * 16 + lane / 4 of both planes.  The LDS image of a piece is wave-uniform base + lane * 16 B; the 16-byte slot swizzle
    // goes on the SOURCE address.  Per lane: three 32-bit byte offsets from wave-uniform plane pointers that advance by one
    // 32-channel chunk (A: in_ld rows, W: wrows rows) per k-step.
    const unsigned ls16 = (unsigned)(((lane & 3) ^ swz64(lane >> 2)) << 4);       // swz64(16 k + row) == swz64(row)
    unsigned aoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        long long m = m0 + (i * 8 + wave) * 16 + (lane >> 2);
        if (m >= a.M) m = a.M - 1;                                  // tail rows: valid address, result never stored
        aoff[i] = (unsigned)m * 64u + ls16;
    }
    unsigned wsoff;
    {
        int n = n0 + wave * 16 + (lane >> 2);
        if (n >= a.wrows) n = a.wrows - 1;                          // columns past the packed rows: never stored
        wsoff = (unsigned)n * 64u + ls16;
    }
    const unsigned char* pa_hi = reinterpret_cast<const unsigned char*>(a.a_hi);
    const unsigned char* pa_lo = reinterpret_cast<const unsigned char*>(a.a_lo);
    const unsigned char* pw_hi = reinterpret_cast<const unsigned char*>(a.w_hi);
    const unsigned char* pw_lo = reinterpret_cast<const unsigned char*>(a.w_lo);
    const long long a_step = (long long)a.in_ld * 64, w_step = (long long)a.wrows * 64;
    int wr_off = 0;                                                 // stage buffer (byte offset) the next issue goes to
    auto issue_stage = [&]() {
        unsigned char* dst = smem + wr_off + wave * 1024;
        dma16(pa_hi + aoff[0], dst);
        dma16(pa_hi + aoff[1], dst + 8 * 1024);
        dma16(pa_lo + aoff[0], dst + A_LO);
        dma16(pa_lo + aoff[1], dst + A_LO + 8 * 1024);
        dma16(pw_hi + wsoff, dst + W_HI);
        dma16(pw_lo + wsoff, dst + W_LO);
        pa_hi += a_step;
        pa_lo += a_step;
        pw_hi += w_step;
        pw_lo += w_step;
        wr_off = wr_off == 2 * STAGE ? 0 : wr_off + STAGE;
    };

    f32x4 acc[4][4], cor[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            cor[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }

#ifdef ATMVFI_STAMP
    unsigned long long tstamp[4], rstamp[4];
#endif
    PP_STAMP(0);
    const int nk = a.nchunks32;
    // ---- prologue: per-tile constants (bias / slope of the column block, and the tile's row-map entries: a global load of the map
    // in the epilogue would cost a memory latency per tile), stages 0 and 1
    atmvfi::gemm_dma_consts<BN>(a, n0, cst, wave, lane);
    if (a.out_row_map && a.mode == ATMVFI_GEMM_LINEAR) {
        long long m = m0 + (wave & 3) * 64 + lane;
        if (m >= a.M) m = a.M - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.out_row_map + m),
                                         (__attribute__((address_space(3))) void*)(cst + atmvfi::gemm_const_floats(BN) + (wave & 3) * 64), 4, 0, 0);
    }
    issue_stage();
    if (nk > 1) {
        issue_stage();
        wait_vm<PIECES>();
    } else {
        wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();          // the second group runs one phase behind
    PP_STAMP(1);

    f16x8 xh[4], xl[4], wh[4], wl[4];
    const unsigned xfrag = lds_offset(smem) + (unsigned)((64 * wm + r) * 64 + ((g ^ swz64(r)) << 4));
    const unsigned wfrag = lds_offset(smem) + (unsigned)(W_HI + (64 * wn + r) * 64 + ((g ^ swz64(r)) << 4));
    int rd_off = 0;                                      // stage buffer (byte offset) of the k-step being read

    // One k-step of one wave.  ISSUE: put the stage two k-steps ahead in flight; WAIT: vmcnt to wait for before the next stage is
    // read (-1: there is no next stage).
    auto kstep = [&](auto issue_c, auto wait_c) {
        constexpr bool ISSUE = decltype(issue_c)::value;
        constexpr int WAIT = decltype(wait_c)::value;
        // ---------------- read phase ----------------
        const unsigned xa = xfrag + (unsigned)rd_off, wa = wfrag + (unsigned)rd_off;
        static_for<0, 4>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            lds_read16<i * 1024>(xh[i], xa);
            lds_read16<i * 1024 + A_LO>(xl[i], xa);
        });
        static_for<0, 4>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            lds_read16<j * 1024>(wh[j], wa);
            lds_read16<j * 1024 + BN * 64>(wl[j], wa);
        });
        rd_off = rd_off == 2 * STAGE ? 0 : rd_off + STAGE;
        if constexpr (ISSUE) issue_stage();
        if constexpr (WAIT >= 0) {
            if (grp == 1) wait_vm<WAIT>();
        }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(xh[0]), "+v"(xh[1]), "+v"(xh[2]), "+v"(xh[3]), "+v"(xl[0]), "+v"(xl[1]), "+v"(xl[2]), "+v"(xl[3]));
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(wh[j]), "+v"(wl[j]));
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- MFMA phase ----------------
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                cor[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[j], xh[i], cor[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xh[i], acc[i][j], 0, 0, 0);
            }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) cor[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xl[i], cor[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if constexpr (WAIT >= 0) {
            if (grp == 0) wait_vm<WAIT>();
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int kc = 0; kc + 2 < nk; ++kc) kstep(std::true_type{}, std::integral_constant<int, PIECES>{});
    if (nk >= 2) kstep(std::false_type{}, std::integral_constant<int, 0>{});
    kstep(std::false_type{}, std::integral_constant<int, -1>{});
    if (grp == 0) __builtin_amdgcn_s_barrier();          // same number of barriers for both groups
    PP_STAMP(2);

    // ---- epilogue.  Lane (r, g) holds, of its 16 MFMA tiles (i, j), GEMM row 64 wm + 16 i + r and columns 64 wn + 16 j + 4 g .. + 3.
    // Correction accumulators folded in first (their registers then hold the residual batch); the row-map entries come from LDS;
    // (LINEAR with a residual whose width is a multiple of 4: every case of the network) ALL sixteen residual vectors in one batch
    // of unconditional loads -- dead rows and columns read the residual's first vector -- so there is one wait per tile and it comes
    // before the first store (vmcnt counts stores on gfx9: a wait per row group would drain the previous group's stores).
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[i][j] = acc[i][j] + cor[i][j] * LO_UNSCALE;
            asm volatile("" : "+v"(acc[i][j]));
        }
    const bool mapped = a.out_row_map && a.mode == ATMVFI_GEMM_LINEAR;
    int ro[4];                 // output row of each group's row: the row map's entry or (unmapped) 0; < 0: nothing to store
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long m = m0 + 64 * wm + 16 * i + r;
        ro[i] = m < a.M ? 0 : -1;
        if (mapped) {
            const int e = reinterpret_cast<const int*>(cst + atmvfi::gemm_const_floats(BN))[64 * wm + 16 * i + r];
            ro[i] = m < a.M ? e : -1;
        }
#ifdef ATMVFI_ABLATE
        if ((a.dbg & 1) && acc[i][0].x != 12345.678f) ro[i] = -1;
#endif
    }
    const bool vec_res = a.residual && (a.Cout & 3) == 0 && a.mode != ATMVFI_GEMM_DECONV;
    f32x4 res[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) res[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (vec_res) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long m = m0 + 64 * wm + 16 * i + r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int nb = n0 + 64 * wn + 16 * j + 4 * g;
                const float* p = (ro[i] >= 0 && nb < a.Cout) ? a.residual + m * (long long)a.res_ld + nb : a.residual;
                res[i][j] = *reinterpret_cast<const f32x4*>(p);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(res[i][j]));
    }
    // (two copies of the store loop under a uniform branch: with the ragged-width residual loads as a conditional inside one loop,
    // hipcc puts their s_waitcnt vmcnt(0) into the shared block, i.e. in front of every store group of the common case too)
    auto store_rows = [&](auto ragged_tag) {
        constexpr bool RAGGED = decltype(ragged_tag)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (ro[i] >= 0) {
                const long long m = m0 + 64 * wm + 16 * i + r;
                float* orow;
                long long prow;
                int pc0;
                atmvfi::gemm_out_row_at(a, m, mapped ? (long long)ro[i] : m, orow, prow, pc0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cl = 64 * wn + 16 * j + 4 * g;
                    const atmvfi::ChanPos cp = atmvfi::gemm_chan_pos(a, n0 + cl);
                    const f32x4 b = *reinterpret_cast<const f32x4*>(cst + cl);
                    const f32x4 p = *reinterpret_cast<const f32x4*>(cst + BN + cl);
                    f32x4 rv = res[i][j];
                    if constexpr (RAGGED) rv = atmvfi::gemm_load_residual4(a.residual + m * (long long)a.res_ld, cp);
                    atmvfi::gemm_finish_store4(a, orow, prow, pc0, cp, acc[i][j], b, p, rv);
                }
            }
        }
    };
    if (a.residual && !vec_res) store_rows(std::true_type{});
    else store_rows(std::false_type{});
#ifdef ATMVFI_STAMP
    PP_STAMP(3);
    if (a.stamp && lane == 0) {
        unsigned long long* o = a.stamp + ((long long)blockIdx.x * 8 + wave) * 8;
        for (int k = 0; k < 3; ++k) { o[k] = tstamp[k + 1] - tstamp[k]; o[4 + k] = rstamp[k + 1] - rstamp[k]; }
        o[3] = (unsigned long long)nk;
    }
#endif
}

}  // namespace

int atmvfi::launch_gemm_pp(const GemmDev& d, int ngemm, hipStream_t s) {
    const size_t lds = (size_t)3 * STAGE + CST_FLOATS * sizeof(float);
    const hipError_t attr_err = atmvfi::allow_dynamic_lds<gemm_pp_kernel>(lds);
    ATMVFI_REQUIRE(attr_err == hipSuccess, ATMVFI_ELAUNCH, "gemm_pp: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    ATMVFI_REQUIRE((long long)d.in_ld * 64 < (1ll << 32) && (long long)d.wrows * 64 < (1ll << 32), ATMVFI_EINVAL,
                   "gemm_pp: plane rows x 64 bytes must fit 32 bits (got %d rows)", d.in_ld);
    GemmDev dd = d;
    dd.dbg = 0;
#ifdef ATMVFI_STAMP
    dd.stamp = g_pp_stamp;
#endif
    dd.nblocks = (ngemm + BN - 1) / BN;
    const long long mgroups = (atmvfi::ceil_div64(d.M, BM) + 7) / 8;
    ATMVFI_REQUIRE(mgroups * 8 * dd.nblocks < (1LL << 31), ATMVFI_EINVAL, "gemm_pp: grid too large");
    hipLaunchKernelGGL(gemm_pp_kernel, dim3((unsigned)(mgroups * 8 * dd.nblocks)), dim3(512), lds, s, dd);
    return atmvfi::check_launch("gemm_pp");
}

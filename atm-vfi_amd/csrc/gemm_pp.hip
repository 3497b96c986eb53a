// LDS-DMA GEMM on split-plane activations, "ping-pong" schedule (gfx950): nn.Linear rows, ConvTranspose2d 2x2 / stride 2 and
// strided / dilated / 1x1 Conv2d (attention.py:138-141, 349-351, 93-96; network_base.py:20-32, 73-85, 417-424) whose input the
// producing layer left as split planes.
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
//   * the k-loop body has no data-dependent branch; the last two k-steps of a tile are separate copies: after the last tile they
//     issue nothing (nothing is in flight when the kernel ends), before that they put the next tile's constants and stages 0-1 in
//     flight -- the grid is PERSISTENT, one workgroup per CU walking its XCD's tiles, and the DMA ring flows across tiles;
//   * the epilogue is transposed through LDS (the wave's own DMA slots of the stage buffer that was read last), so that the 16 lanes
//     of an output row cover 256 contiguous bytes; both groups run it together (the first waits a phase for the second).
// Details and measurements at the places in the code and in DESIGN.md section 3.0b.
#include "common.h"
#include "gemm_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#ifdef ATMVFI_STAMP
static unsigned long long* g_pp_stamp = nullptr;
extern "C" void atmvfi_debug_set_pp_stamp_buffer(void* p) { g_pp_stamp = (unsigned long long*)p; }
// per-k-step cycle sums (k-step index inside its tile, 0..15): 16 more values per wave behind the 8 of the phase stamps of every wave
#define PP_KSTAMP_BEGIN() unsigned long long kt0_ = 0; do { if (a.stamp) { __builtin_amdgcn_sched_barrier(0); kt0_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define PP_KSTAMP_END(idx) do { if (a.stamp) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) atomicAdd(&kst[wave * 16 + ((idx) < 15 ? (idx) : 15)], t_ - kt0_); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define PP_STAMP(i) do { if (a.stamp) { tstamp[i] = __builtin_amdgcn_s_memtime(); rstamp[i] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#define PP_SUB(k) do { if (a.stamp) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); sub[k] += t_ - sub_t; sub_t = t_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define PP_STAMP(i) do { } while (0)
#define PP_SUB(k) do { } while (0)
#define PP_KSTAMP_BEGIN() do { } while (0)
#define PP_KSTAMP_END(idx) do { } while (0)
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
__device__ __forceinline__ void lds_write16f(unsigned addr, const f32x4& v) {
    asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_read16f(f32x4& d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <bool CONVM>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(const GemmDev a) {
    fp16_saturate_on();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int r = lane & 15;
    const int g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;

    // PERSISTENT GRID over an XCD-aware tile order.  Virtual block v: xcd = v & 7, slot = v >> 3, row tile = (slot / nblocks) * 8 + xcd,
    // column block = slot % nblocks (blocks v and v+8 share an XCD, so the column blocks of one row tile go to one L2).  Workgroup b
    // walks v = b, b + grid, ... (grid a multiple of 8: it stays on its XCD) and the DMA ring KEEPS FLOWING ACROSS TILES: the last two
    // k-steps of a tile put stages 0 and 1 of the workgroup's next tile in flight, so a tile's first DMA round trip (7-8 k cycles of
    // a 42 k-cycle K = 384 tile, tools/stamp_split.py) runs under the previous tile's last k-steps and epilogue.
    // (The launcher makes the grid persistent only for nk >= 2: the look-ahead of two stages then spans one tile boundary at most.)
    const int grid = gridDim.x;
    auto tile_of = [&](int v, int& tm0, int& tn0) {
        const int xcd = v & 7;
        const int slot = v >> 3;
        const int mgrp = slot / a.nblocks;
        const int nblk = slot - mgrp * a.nblocks;
        tm0 = (mgrp * 8 + xcd) * BM;
        tn0 = nblk * BN;
    };
    const int M = (int)a.M;                    // rows < 2^26 (launcher): 32-bit row arithmetic throughout
    int vb = blockIdx.x;
    int m0;
    int n0;
    tile_of(vb, m0, n0);
    if (m0 >= M) return;
    float* cst_base = reinterpret_cast<float*>(smem + 3 * STAGE);       // two buffers of per-tile constants

    // ---- DMA pieces of this wave.  A: rows (i * 8 + wave) * 16 + lane / 4 (i = 0, 1) of the hi and of the lo plane; W: rows
    // wave * 16 + lane / 4 of both planes.  The LDS image of a piece is wave-uniform base + lane * 16 B; the 16-byte slot swizzle
    // goes on the SOURCE address.  Per lane: three 32-bit byte offsets from wave-uniform plane pointers that advance by one
    // 32-channel chunk (A: in_ld rows, W: wrows rows) per k-step.  They belong to the ISSUER's tile: the tile whose stages are being
    // put in flight -- the tile of the MFMAs or, in its last two k-steps, the next one.
    const unsigned ls16 = (unsigned)(((lane & 3) ^ swz64(lane >> 2)) << 4);       // swz64(16 k + row) == swz64(row)
    unsigned aoff[2];
    unsigned wsoff;
    const unsigned char* pa_hi;
    const unsigned char* pa_lo;
    const unsigned char* pw_hi;
    const unsigned char* pw_lo;
    // CONV mode (strided / dilated / 1x1 convolutions, network_base.py:20-25, 73-85): k-step = (tap, 32-channel chunk); GEMM row m is
    // output pixel (n, oy, ox) and its operand row at tap (ky, kx) is input pixel (n, oy*stride - pad + ky*dil, ox*stride - pad +
    // kx*dil), or the planes' zero row N*H*W when that falls outside the image -- the same DMA with a per-lane source row.  Per
    // lane and row: the input row of tap (0, 0) and its (y, x) packed into one register; wave-uniform: tap and chunk of the
    // stage that goes out next (aoff is not used).
    int crow[2], cyx[2];
    int c_ky = 0, c_kx = 0, c_chunk = 0;
    const int zero_row = a.in_N * a.H * a.W;
    auto setup_issue = [&](int tm0, int tn0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int m = tm0 + (i * 8 + wave) * 16 + (lane >> 2);
            if (m >= M) m = M - 1;                                  // tail rows: valid address, result never stored
            if constexpr (CONVM) {
                // (divisors laundered through an empty asm: otherwise hipcc hoists their reciprocals out of the tile loop and keeps
                // them in vector registers across the k-loop, which is at the 256-register limit)
                int hw = a.Ho * a.Wo, wo = a.Wo;
                asm volatile("" : "+s"(hw), "+s"(wo));
                const int n = (int)((unsigned)m / (unsigned)hw);
                const int rem = m - n * hw;
                const int oy = rem / wo, ox = rem - oy * wo;
                const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
                cyx[i] = (iy0 << 16) | (ix0 & 0xffff);
                crow[i] = (n * a.H + iy0) * a.W + ix0;
            } else {
                aoff[i] = (unsigned)m * 64u + ls16;
            }
        }
        int n = tn0 + wave * 16 + (lane >> 2);
        if (n >= a.wrows) n = a.wrows - 1;                          // columns past the packed rows: never stored
        wsoff = (unsigned)n * 64u + ls16;
        pa_hi = reinterpret_cast<const unsigned char*>(a.a_hi);
        pa_lo = reinterpret_cast<const unsigned char*>(a.a_lo);
        pw_hi = reinterpret_cast<const unsigned char*>(a.w_hi);
        pw_lo = reinterpret_cast<const unsigned char*>(a.w_lo);
        c_ky = 0;
        c_kx = 0;
        c_chunk = 0;
    };
    setup_issue(m0, n0);
    const long long a_step = (long long)a.in_ld * 64, w_step = (long long)a.wrows * 64;
    int wr_off = 0;                                                 // stage buffer (byte offset) the next issue goes to
    auto issue_stage = [&]() {
        unsigned char* dst = smem + wr_off + wave * 1024;
        if constexpr (CONVM) {
            // plane pointers of this stage's chunk: the first source, or (channels >= 32 * split_chunks) the second one
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
            dma16(bh + off[1], dst + 8 * 1024);
            dma16(bl + off[0], dst + A_LO);
            dma16(bl + off[1], dst + A_LO + 8 * 1024);
            if (++c_chunk == a.cpt32) {
                c_chunk = 0;
                if (++c_kx == a.kw) { c_kx = 0; ++c_ky; }
            }
        } else {
            dma16(pa_hi + aoff[0], dst);
            dma16(pa_hi + aoff[1], dst + 8 * 1024);
            dma16(pa_lo + aoff[0], dst + A_LO);
            dma16(pa_lo + aoff[1], dst + A_LO + 8 * 1024);
            pa_hi += a_step;
            pa_lo += a_step;
        }
        dma16(pw_hi + wsoff, dst + W_HI);
        dma16(pw_lo + wsoff, dst + W_LO);
        pw_hi += w_step;
        pw_lo += w_step;
        wr_off = wr_off == 2 * STAGE ? 0 : wr_off + STAGE;
    };
    // Per-tile constants by LDS-DMA, two pieces per wave: bias / slope of the column block, and the tile's BM row-map entries (LINEAR
    // with an out_row_map: the window-reverse scatter of proj; a global load of the map in the epilogue would cost a memory latency
    // per tile).  Without a map the second piece repeats the first (same bytes, same place): the counted waits stay uniform.
    auto dma_tile_consts = [&](int tm0, int tn0, float* dst) {
        atmvfi::gemm_dma_consts<BN>(a, tn0, dst, wave, lane);
        if (a.out_row_map && a.mode == ATMVFI_GEMM_LINEAR) {
            int m = tm0 + (wave & 3) * 64 + lane;
            if (m >= M) m = M - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.out_row_map + m),
                                             (__attribute__((address_space(3))) void*)(dst + atmvfi::gemm_const_floats(BN) + (wave & 3) * 64), 4, 0, 0);
        } else {
            atmvfi::gemm_dma_consts<BN>(a, tn0, dst, wave, lane);
        }
    };

    f32x4 acc[4][4], cor[4][4];

#ifdef ATMVFI_STAMP
    unsigned long long* kst = reinterpret_cast<unsigned long long*>(smem + 3 * STAGE + 2 * CST_FLOATS * sizeof(float));
    if (tid < 128) kst[tid] = 0;
    __syncthreads();
    unsigned long long tstamp[4], rstamp[4];
    unsigned long long t_loop = 0, t_epi = 0, r_loop = 0, r_epi = 0, t_pro = 0, r_pro = 0;
    int ntile = 0;
    unsigned long long sub[3] = {0, 0, 0}, sub_t = 0;
#endif
    PP_STAMP(0);
    const int nk = a.nchunks32;
    // ---- prologue of the first tile: constants, stages 0 and 1
    dma_tile_consts(m0, n0, cst_base);
    issue_stage();
    if (nk > 1) {
        issue_stage();
        wait_vm<PIECES>();
    } else {
        wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();          // the second group runs one phase behind

    f16x8 xh[4], xl[4], wh[4], wl[4];
    const unsigned xfrag = lds_offset(smem) + (unsigned)((64 * wm + r) * 64 + ((g ^ swz64(r)) << 4));
    const unsigned wfrag = lds_offset(smem) + (unsigned)(W_HI + (64 * wn + r) * 64 + ((g ^ swz64(r)) << 4));
    int rd_off = 0;                                      // stage buffer (byte offset) of the k-step being read
    int seq = 0;                                         // tiles done by this workgroup
    // this workgroup's next tile (row tiles only grow along a workgroup's walk: the first empty one ends it)
    int nm0 = 0;
    int nn0 = 0;
    bool has_next = false;

    // One k-step of one wave.  KIND 0: a k-step with two more k-steps of its tile behind it -- the stage two k-steps ahead goes out,
    // six pieces stay in flight at the wait.  KIND 1: the tile's last k-step but one -- the ring moves on to the next tile (its
    // constants and stage 0; eight pieces in flight), or nothing goes out (wait for everything).  KIND 2: the last k-step --
    // stage 1 of the next tile, or nothing (and no wait: there is no next stage).  g1wait: false in the first k-step of a later
    // tile for the second group, which waited before its epilogue (its stores would otherwise sit in front of the counted wait).
    auto kstep = [&](auto kind_c, bool g1wait) {
        constexpr int KIND = decltype(kind_c)::value;
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
        if constexpr (KIND == 0) {
            issue_stage();
        } else if constexpr (KIND == 1) {
            if (has_next) {
                setup_issue(nm0, nn0);
                dma_tile_consts(nm0, nn0, cst_base + ((seq + 1) & 1) * CST_FLOATS);
                issue_stage();
            }
        } else {
            if (has_next) issue_stage();
        }
        auto wait_next = [&]() {
            if constexpr (KIND == 0) {
                wait_vm<PIECES>();
            } else if constexpr (KIND == 1) {
                if (has_next) wait_vm<PIECES + 2>();
                else wait_vm<0>();
            } else {
                if (has_next) wait_vm<PIECES>();
            }
        };
        if (grp == 1 && g1wait) wait_next();
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
        if (grp == 0) wait_next();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    for (;;) {
        const int nxt = vb + grid;
        has_next = false;
        if (nxt < a.vblocks) {
            tile_of(nxt, nm0, nn0);
            has_next = nm0 < M;
        }
        const float* cst = cst_base + (seq & 1) * CST_FLOATS;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                cor[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                // pinned: hipcc otherwise folds the zeros into the first MFMAs' C operand and uses the (then dead) accumulator
                // registers as temporaries of the first read phase, guarded by s_waitcnt vmcnt(0)
                asm volatile("" : "+v"(acc[i][j]), "+v"(cor[i][j]));
            }
        PP_STAMP(1);
        {
            bool g1wait = seq == 0;
            for (int kc = 0; kc + 2 < nk; ++kc) {
                PP_KSTAMP_BEGIN();
                kstep(std::integral_constant<int, 0>{}, g1wait);
                PP_KSTAMP_END(kc);
                g1wait = true;
            }
            if (nk >= 2) {
                PP_KSTAMP_BEGIN();
                kstep(std::integral_constant<int, 1>{}, g1wait);
                PP_KSTAMP_END(nk - 2);
                g1wait = true;
            }
            {
                PP_KSTAMP_BEGIN();
                kstep(std::integral_constant<int, 2>{}, g1wait);
                PP_KSTAMP_END(nk - 1);
            }
        }
        PP_STAMP(2);
#ifdef ATMVFI_STAMP
        sub_t = __builtin_amdgcn_s_memtime();
#endif
        // Both groups run the epilogue AT THE SAME TIME: the first group waits here for the second one's last MFMA phase (one phase,
        // ~900 cycles), and two epilogue waves per SIMD fill each other's VALU latencies.  (One group's epilogue beside the other
        // group's last MFMA phase / first read phase of the next tile, i.e. one after the other, cost 15 k cycles per tile boundary
        // against 22 k of k-loop at K = 384: a lone epilogue wave per SIMD runs its dependent VALU chains at ~7 cycles an instruction.)
        if (grp == 0) __builtin_amdgcn_s_barrier();
        // The second group's pieces of the next tile's stage 1 (all that is in flight) are waited for HERE, before its stores join
        // the queue; its first k-step of the next tile then skips the counted wait (g1wait).
        if (grp == 1 && has_next) wait_vm<0>();

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
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = acc[i][j] + cor[i][j] * LO_UNSCALE;
                asm volatile("" : "+v"(acc[i][j]));
            }
        {
            const unsigned xb = (unsigned)(rd_off == 0 ? 2 * STAGE : rd_off - STAGE);          // buffer of the last k-step
            const unsigned tb = lds_offset(smem) + xb + (unsigned)(wave * 1024);
            const unsigned wbase = tb + (unsigned)((r >> 2) * 8192 + (r & 3) * 256 + ((g ^ (r & 3)) << 4));
            const unsigned rq6 = (unsigned)((r >> 2) << 6);
            const unsigned rbase = tb + (unsigned)(g * 256);
            const unsigned rg4 = (unsigned)((r ^ g) << 4);
            unsigned wad[4], rad[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                wad[k] = wbase + (rq6 ^ (unsigned)(k << 6));                    // slot (4 j + g) ^ r of row r
                rad[k] = rbase + (rg4 ^ (unsigned)(k << 6));                    // slot r ^ (4 q + g) of row 4 q + g (+ q * 8192 below)
            }
            static_for<0, 4>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                static_for<0, 4>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    lds_write16f(wad[j], acc[i][j]);
                });
                static_for<0, 4>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    lds_read16f<q * 8192>(acc[i][q], rad[q]);
                });
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(acc[i][q]));
        }
        PP_SUB(0);
        const bool mapped = a.out_row_map && a.mode == ATMVFI_GEMM_LINEAR;
        const int cl = 64 * wn + 4 * r;                          // the lane's four columns inside the column block
        const int nb = n0 + cl;
        const atmvfi::ChanPos cp = atmvfi::gemm_chan_pos(a, nb);
        const f32x4 bvec = *reinterpret_cast<const f32x4*>(cst + cl);
        const f32x4 pvec = *reinterpret_cast<const f32x4*>(cst + BN + cl);
        const int mrow = m0 + 64 * wm + g;                       // row of (i, q) = (0, 0); (i, q) adds 16 i + 4 q
        int ro[4][4];              // output row: the row map's entry, or (unmapped) the row itself; < 0: nothing to store
        if (mapped) {                 // (the test outside the loops: inside, hipcc makes it a scalar branch per row)
            const int* mp = reinterpret_cast<const int*>(cst + atmvfi::gemm_const_floats(BN)) + 64 * wm + g;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) ro[i][q] = (mrow + 16 * i + 4 * q) < M ? mp[16 * i + 4 * q] : -1;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = mrow + 16 * i + 4 * q;
                    ro[i][q] = m < M ? m : -1;
                }
        }
        const bool vec_res = !CONVM && a.residual && (a.Cout & 3) == 0 && a.mode != ATMVFI_GEMM_DECONV;
        f32x4 res[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) res[i][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (vec_res && a.fit32 && m0 + BM <= M) {
            // (all rows live: 32-bit offsets from a scalar base, stepped by 4 rows)
            const unsigned char* rb = reinterpret_cast<const unsigned char*>(a.residual);
            unsigned voff = nb < a.Cout ? ((unsigned)mrow * (unsigned)a.res_ld + (unsigned)nb) * 4u : 0u;
            const unsigned step = nb < a.Cout ? (unsigned)a.res_ld * 16u : 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    res[i][q] = *reinterpret_cast<const f32x4*>(rb + voff);
                    voff += step;
                }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(res[i][q]));
        } else if (vec_res) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = mrow + 16 * i + 4 * q;
                    const float* p = (ro[i][q] >= 0 && nb < a.Cout) ? a.residual + m * (long long)a.res_ld + nb : a.residual;
                    res[i][q] = *reinterpret_cast<const f32x4*>(p);
                }
            // every vector is "used" here, dead rows' too: a load left pending on some path makes hipcc guard the next tile's first
            // write to its register with s_waitcnt vmcnt(0), which would also wait for this tile's stores
#pragma unroll
            for (int i = 0; i < 4; ++i)
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
            for (int i = 0; i < 4; ++i)
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
        PP_SUB(1);
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
                    for (int i = 0; i < 4; ++i)
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
                for (int i = 0; i < 4; ++i)
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
                    for (int i = 0; i < 4; ++i)
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
                    for (int i = 0; i < 4; ++i)
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
                    for (int i = 0; i < 4; ++i)
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
                for (int i = 0; i < 4; ++i)
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
        PP_SUB(2);
        PP_STAMP(3);
#ifdef ATMVFI_STAMP
        if (a.stamp) {
            if (seq == 0) { t_pro = tstamp[1] - tstamp[0]; r_pro = rstamp[1] - rstamp[0]; }
            t_loop += tstamp[2] - tstamp[1]; r_loop += rstamp[2] - rstamp[1];
            t_epi += tstamp[3] - tstamp[2]; r_epi += rstamp[3] - rstamp[2];
            ++ntile;
        }
#endif
        if (!has_next) break;
        if (grp == 1) __builtin_amdgcn_s_barrier();          // the second group drops one phase behind again
        vb = nxt;
        m0 = nm0;
        n0 = nn0;
        ++seq;
    }
#ifdef ATMVFI_STAMP
    if (a.stamp && lane == 0) {
        unsigned long long* o = a.stamp + ((long long)blockIdx.x * 8 + wave) * 8;
        o[0] = t_pro; o[1] = t_loop / ntile; o[2] = t_epi / ntile;
        o[3] = (unsigned long long)nk;
        o[4] = r_loop / ntile; o[5] = sub[0] / ntile; o[6] = sub[1] / ntile;
        o[7] = sub[2] / ntile;
    }
    __syncthreads();
    if (a.stamp && tid < 128) a.stamp[(long long)gridDim.x * 64 + (long long)blockIdx.x * 128 + tid] = kst[tid] / (ntile ? ntile : 1);
#endif
}

}  // namespace

template <bool CONVM>
static int launch_pp(const GemmDev& d, int ngemm, hipStream_t s) {
#ifdef ATMVFI_STAMP
    const size_t lds = (size_t)3 * STAGE + 2 * CST_FLOATS * sizeof(float) + 1024;      // + the per-k-step stamp sums
#else
    const size_t lds = (size_t)3 * STAGE + 2 * CST_FLOATS * sizeof(float);
#endif
    const hipError_t attr_err = atmvfi::allow_dynamic_lds<gemm_pp_kernel<CONVM>>(lds);
    ATMVFI_REQUIRE(attr_err == hipSuccess, ATMVFI_ELAUNCH, "gemm_pp: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    ATMVFI_REQUIRE((long long)d.in_ld * 64 < (1ll << 32) && (long long)d.wrows * 64 < (1ll << 32) &&
                       (!d.a_hi2 || (long long)d.in_ld2 * 64 < (1ll << 32)) && d.M < (1ll << 26), ATMVFI_EINVAL,
                   "gemm_pp: plane rows x 64 bytes must fit 32 bits (got %d rows)", d.in_ld);
    if (CONVM) ATMVFI_REQUIRE(d.H < 32768 && d.W < 32768, ATMVFI_EINVAL, "gemm_pp: CONV mode packs (y, x) into 16 bits each");
    GemmDev dd = d;
    dd.dbg = 0;
    dd.pfit32 = d.out_hi && (long long)d.out_plane_rows * 64 < (1ll << 32);
    dd.fit32 = d.out && !d.out_row_map && d.out_rpg == 0 && d.mode != ATMVFI_GEMM_DECONV && (d.M + 1) * (long long)d.out_ld * 4 < (1ll << 32) &&
               (!d.residual || (d.M + 1) * (long long)d.res_ld * 4 < (1ll << 32));
#ifdef ATMVFI_STAMP
    dd.stamp = g_pp_stamp;
#endif
    dd.nblocks = (ngemm + BN - 1) / BN;
    const long long mgroups = (atmvfi::ceil_div64(d.M, BM) + 7) / 8;
    ATMVFI_REQUIRE(mgroups * 8 * dd.nblocks < (1LL << 31), ATMVFI_EINVAL, "gemm_pp: grid too large");
    dd.vblocks = (int)(mgroups * 8 * dd.nblocks);
    // persistent from two k-steps up (the ring's look-ahead of two stages then spans at most one tile boundary): one workgroup per
    // CU (147 KiB of LDS) walking its XCD's tiles; K <= 32: one workgroup per tile
    const int grid = d.nchunks32 >= 2 ? std::min(dd.vblocks, atmvfi::cu_count()) : dd.vblocks;
    hipLaunchKernelGGL(gemm_pp_kernel<CONVM>, dim3((unsigned)grid), dim3(512), lds, s, dd);
    return atmvfi::check_launch("gemm_pp");
}

int atmvfi::launch_gemm_pp(const GemmDev& d, int ngemm, hipStream_t s) {
    return d.mode == ATMVFI_GEMM_CONV ? launch_pp<true>(d, ngemm, s) : launch_pp<false>(d, ngemm, s);
}

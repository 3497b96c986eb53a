// 3x3 / stride 1 / pad 1 split-precision convolution on SPLIT-PLANE activations, "ping-pong" schedule (gfx950).
//
// Same arithmetic as conv3x3_f16x3_row.hip (x = hi + lo'/1024 in fp16, three v_mfma_f32_16x16x32_f16 per product into two fp32
// accumulators, identical k order: bit-identical results), same weight planes, same 16x16-pixel x 16*WN-channel tile on 512
// threads -- but nothing is staged through registers and no two waves of a SIMD do the same thing at the same time:
//
//   * the input is read as the split planes (hi, lo', chunk major [32-channel chunk][pixel][32], common.h RowSink) that the
//     PRODUCING layer wrote in its epilogue, so the (16+2)x(16+2) halo of a chunk goes global -> LDS by LDS-DMA
//     (global_load_lds_dwordx4: no VGPR, no VALU split, no ds_write); pixels outside the image read a page of zeros.  The halo
//     is double-buffered: chunk c+1 lands while chunk c is consumed;
//   * the weights of one k-step (one tap x 32 channels) are one slot of an LDS ring of 4-6 slots (whatever fits beside the two
//     halo buffers), filled ring - 1 k-steps ahead;
//   * the eight waves form two groups (waves 0-3 / 4-7 = the two waves of each SIMD) that run ONE PHASE APART: while a group
//     issues the 6*WN MFMAs of k-step u from registers, its SIMD partners read the fragments of their next k-step from LDS and
//     issue their share of the DMA; a raw s_barrier swaps the roles.  The matrix pipe of a SIMD is therefore always fed by one of
//     its two waves, and LDS reads / DMA issue / waits never sit in series with the MFMAs (the row kernel's measured bound:
//     matrix pipes 47 % busy with every other unit below 25 %, DESIGN.md section 3.1 "Phases in series").
//
// Ordering rules (cdna_hip_programming.md "Read a staged buffer one phase AFTER the wait that retires it"):
//   * a wave waits for its own DMA pieces of k-step u+1 (counted vmcnt: later pieces stay in flight) and for its own fragment
//     reads of k-step u (lgkmcnt(0)) at the END of its read phase, before the barrier; the group that reads k-step u+1 first
//     does so after the barrier that follows the later group's wait;
//   * a ring slot / halo buffer is refilled by a DMA issued at least one barrier after its last reader's lgkmcnt(0).
// Every wave issues the same number of DMA instructions in every phase (a wave without a piece of its own repeats one), and the
// k-loop has no branch: the vmcnt waits are immediates.
#include "conv3_common.h"

#include <stdlib.h>

#ifdef ATMVFI_STAMP
// Diagnostic build only (`make stamp`, tools/stamp_conv3p.py): per-wave s_memtime sums of the phases of a k-step.
static unsigned long long* g_planes_stamp = nullptr;
extern "C" void atmvfi_debug_set_planes_stamp_buffer(void* p) { g_planes_stamp = (unsigned long long*)p; }
#ifdef ATMVFI_STAMP_KSTEP
#define PDBG(bit) ((a.dbg & (bit)) != 0)
#else
#define PDBG(bit) false
#endif
#define PSTAMP_DECL unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_epi = 0; unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
#define TSTAMP(k) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); st_[k] += t1_ - st_t0; st_t0 = t1_; __builtin_amdgcn_sched_barrier(0); }
#ifdef ATMVFI_STAMP_KSTEP
// (per-phase stamps inside the k-step: eight s_memtime per k-step and 16 more live scalars -- since the persistent grid this build spills
// and its phase numbers are not the product's; the default stamp build only times whole tiles: TSTAMP below)
#define PSTAMP(k) TSTAMP(k)
#else
#define PSTAMP(k)
#endif
#else
#define PDBG(bit) false
#define PSTAMP_DECL
#define PSTAMP(k)
#define TSTAMP(k)
#endif

namespace {

struct Conv3PDev {
    const _Float16* in_hi;      // input planes, already advanced to the first 32-channel chunk of the view
    const _Float16* in_lo;
    long long in_rows;          // plane rows (> N*H*W; row N*H*W of every chunk is zero): chunk stride = in_rows * 32 halves
    int N, H, W, Cin;
    const _Float16* w_hi;
    const _Float16* w_lo;
    int wrows, cf, tail;
    int Cout;
    float* out;                 // optional fp32 NHWC view
    int out_ld;
    const float* bias;
    const float* prelu;
    _Float16* out_hi;           // optional plane sink (chunk major), channels out_c0 .. out_c0 + Cout
    _Float16* out_lo;
    long long plane_rows;
    int out_c0;
    const float* plane_prelu;   // optional PReLU applied to the plane copy only (the next layer's leading activation)
    _Float16* out_hi2;          // optional SECOND plane sink, raw (no PReLU of its own): a decoder map goes on both through the next
    _Float16* out_lo2;          // stage's leading PReLU (first sink) and as it is (the U-Net's strided convs read it)
    long long plane_rows2;
    int out_c02;
    int out_cmin;               // fp32 output: only channels >= out_cmin (multiple of 4) are stored
    int tiles_x, tiles_y, nblocks, tchunk;
    int vblocks;                // virtual blocks (tiles incl. XCD padding) walked by the persistent grid
    // the tile decode's divisors as multiply-shift pairs (launch_planes: q = (mul_hi(n, m) + n) >> s, exact for n < 2^31): dividing by a
    // kernel argument costs ~18 scalar instructions, and the decode of a workgroup's next-but-one tile -- five divisions -- sits on the
    // critical path of every tile boundary (1.3-1.5 k cycles per tile, tools/stamp_conv3p.py)
    unsigned dm_nblocks, ds_nblocks, dm_perimg, ds_perimg, dm_grp, ds_grp, dm_rows, ds_rows;
    // split-K (under-filled grids with long K, round 4): gridDim.y = ksplit workgroups per tile; split s takes the cps full chunks from
    // chunk s * cps on (the last one the rest and the tap-packed tail) and stores its raw fp32 sums at out + s * part_stride; a second
    // kernel adds the partial sums in split order and runs the epilogue.  1 = off.
    int ksplit, cps;
    long long part_stride;
    // fused read-out (refine_head.0 -> refine_head.1, network_base.py:257-260; WN = 2 or 4, one column block): the tile's activated
    // output -- still in registers, in the accumulator layout, which IS the B-operand layout of the next MFMA -- is multiplied by the
    // 27 x Cout matrix W2[(tap, o)][c] of the following 3-output 3x3 convolution; the 27 per-pixel "tap contributions" go to planar
    // fp32 h2_out[(tap * 3 + o) * h2_plane + pixel] and atmvfi_refine_tail adds each output pixel's nine shifted contributions.
    const _Float16* h2_w;       // [plane hi / lo][row tile 2][k-step WN/2][lane 64][8 halves], k order = this kernel's register order
    float* h2_out;
    long long h2_plane;
    unsigned long long* stamp;  // diagnostic builds only (ATMVFI_STAMP)
    int dbg;                    // diagnostic builds only: ATMVFI_P3_DBG bits switch pieces of the loop off (wrong results, timing only)
};

constexpr int HALO_PIX = HW_ * HW_;           // 324
constexpr int HALO_PLANE_PIECES = (HALO_PIX + 15) / 16;     // 21 one-KiB pieces (16 pixel rows x 64 B) per plane, 12 pad rows
constexpr int HALO_LO = HALO_PLANE_PIECES * 1024;            // byte offset of the lo plane inside a halo buffer
constexpr int HALO_BYTES = 2 * HALO_LO;

// weight ring depth: as many k-steps as fit beside the two halo buffers and the epilogue constants in 160 KiB, at most 5
// epilogue constants of a tile in LDS: bias, PReLU slope and the plane sink's own slope for its BN columns (three rows of BN floats)
constexpr int planes_const_floats(int BN) { return (3 * BN + 63) / 64 * 64; }
constexpr int ring_slots(int wn) {
    const int free_bytes = 160 * 1024 - 2 * HALO_BYTES - 2 * planes_const_floats(16 * wn) * 4;     // two halo buffers, two buffers of epilogue constants
    const int n = free_bytes / (2 * 16 * wn * 64);
    return n > 5 ? 5 : n;            // the static vmcnt counts of the k-loop assume a lookahead of at most 4 k-steps
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void dma16(const unsigned char* src, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// One LDS-DMA instruction per wave (4 bytes per lane): cst[c] = bias[n0 + c], cst[BN + c] = slope[n0 + c], cst[2 BN + c] = plane_slope[n0 + c]
// for the tile's BN columns, 0 / 1 / 1 where an array is absent or the column is past Cout.  (The plane sink's slopes used to be global loads
// inside the store loop: every one of them made hipcc wait for vmcnt(0), i.e. for the previous n-tile pair's STORES to be acknowledged --
// the sink epilogue of a 112-column tile took 9.4 k cycles against 5.1 k for fp32 rows, tools/stamp_conv3p.py.)
template <int BN>
__device__ __forceinline__ void dma_planes_consts(const float* bias, const float* slope, const float* plane_slope, int Cout, int n0, float* cst,
                                                  int wave, int lane) {
    constexpr int NPIECE = (3 * BN + 63) / 64;
    static_assert(NPIECE <= 8, "one piece per wave");
    const int piece = wave % NPIECE;
    const int t = piece * 64 + lane;
    const int row = t / BN;
    const int col = n0 + t - row * BN;
    const float* src = row == 0 ? bias : row == 1 ? slope : plane_slope;
    const float* p = (src && row < 3 && col < Cout) ? src + col : &kEpilogueDefaults[row == 0 ? 0 : 1];
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                     (__attribute__((address_space(3))) void*)(cst + piece * 64), 4, 0, 0);
}

template <int WN>
__global__ __launch_bounds__(512, 1) void conv3x3_planes_kernel(const Conv3PDev a) {
    fp16_saturate_on();
    constexpr int BN = 16 * WN;
    constexpr int WSLOT = 2 * BN * 64;                  // bytes of one ring slot: [hi BN rows][lo BN rows] x 64 B
    constexpr int SW = (2 * WN + 7) / 8;                // weight pieces per wave and k-step (some waves one fewer)
    constexpr int NB = ring_slots(WN);                  // weight ring slots (k-steps)
    constexpr int LA = NB - 1;                          // k-steps between a slot's DMA issue and its first read
    constexpr int CSTF = planes_const_floats(BN);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned halo0 = lds_offset(smem);                        // two halo buffers
    const unsigned ring0 = halo0 + 2 * HALO_BYTES;                  // NB weight slots
    float* cst_base = reinterpret_cast<float*>(smem + 2 * HALO_BYTES + NB * WSLOT);     // two buffers of epilogue constants

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int r = lane & 15;
    const int g = lane >> 4;

    // operands of this workgroup's K range (split-K: blockIdx.y picks the chunk range and the partial-sum buffer; else everything)
    const _Float16* in_hi = a.in_hi;
    const _Float16* in_lo = a.in_lo;
    const _Float16* w_hi = a.w_hi;
    const _Float16* w_lo = a.w_lo;
    float* out_f32 = a.out;
    int nfull = a.cf >> 5;
    int ktail = a.tail;
    if (a.ksplit > 1) {
        // whole chunks, dealt evenly: the first (nfull mod ksplit) ranges take one chunk more; the tap-packed tail goes with the last
        const int sp = blockIdx.y;
        const int rem = nfull - a.cps * a.ksplit;
        const int c0 = sp * a.cps + (sp < rem ? sp : rem);
        const bool last = sp == a.ksplit - 1;
        in_hi += (long long)c0 * a.in_rows * 32;
        in_lo += (long long)c0 * a.in_rows * 32;
        w_hi += (long long)c0 * 9 * a.wrows * 32;
        w_lo += (long long)c0 * 9 * a.wrows * 32;
        nfull = a.cps + (sp < rem ? 1 : 0);
        ktail = last ? ktail : 0;
        out_f32 += sp * a.part_stride;
    }

    // PERSISTENT GRID over the XCD-aware tile order (conv3x3_f16x3_row.hip): virtual block v -> column blocks of one tile back to
    // back on one XCD, each XCD walking a contiguous eighth of the tiles in groups of 8 tile rows, column by column.  Workgroup b
    // walks v = b, b + grid, ... (grid a multiple of 8: it stays on its XCD; the tile index only grows along the walk, so the first
    // empty block ends it).  The DMA streams keep flowing across tiles: the slots that re-sent the last weights / the current halo
    // "into buffers nobody reads" at the end of a tile now carry the NEXT tile's first k-steps and its first halo, so a tile's
    // prologue (first DMA round trip + address arithmetic, 6.5 % of a 101-wide full-resolution tile) runs under the previous
    // tile's last k-steps.
    const int grid = gridDim.x;
    const int per_img = a.tiles_x * a.tiles_y;
    auto fdiv = [](int n, unsigned m, unsigned sh) -> int { return (int)((__umulhi((unsigned)n, m) + (unsigned)n) >> sh); };
    auto decode = [&](int v, int& t_img, int& t_ox0, int& t_oy0, int& t_n0) -> bool {
        int sgrp, nblk, L;
        if (a.ksplit > 1) {
            // split-K launches have fewer tiles than CUs: plain order (tile = v / nblocks), so that the workgroups -- dealt round-robin
            // over the XCDs -- spread over the whole chip instead of filling the first eighths of the XCD-aware order
            sgrp = fdiv(v, a.dm_nblocks, a.ds_nblocks);
            nblk = v - sgrp * a.nblocks;
            L = sgrp;
        } else {
            const int slot = v >> 3;
            sgrp = fdiv(slot, a.dm_nblocks, a.ds_nblocks);
            nblk = slot - sgrp * a.nblocks;
            L = (v & 7) * a.tchunk + sgrp;
        }
        if (v >= a.vblocks || L >= a.N * per_img) return false;
        t_img = fdiv(L, a.dm_perimg, a.ds_perimg);
        L -= t_img * per_img;
        const int tgrp = fdiv(L, a.dm_grp, a.ds_grp);              // groups of 8 tile rows
        const int rem = L - tgrp * 8 * a.tiles_x;
        const bool full = a.tiles_y - 8 * tgrp >= 8;
        const int rows_here = full ? 8 : a.tiles_y - 8 * tgrp;     // (the last group of a map may be shorter: tiles_y mod 8 rows)
        const int txb = full ? rem >> 3 : fdiv(rem, a.dm_rows, a.ds_rows);
        const int tyb = 8 * tgrp + (rem - txb * rows_here);
        t_ox0 = txb * TW;
        t_oy0 = tyb * 16;
        t_n0 = nblk * BN;
        return true;
    };
    int vb = blockIdx.x;
    int img, ox0, oy0, n0;                       // the tile of the MFMAs / of the epilogue
    if (!decode(vb, img, ox0, oy0, n0)) return;
    int nimg = 0, nox0 = 0, noy0 = 0, nn0 = 0;   // this workgroup's next tile
    bool has_next = decode(vb + grid, nimg, nox0, noy0, nn0);

    // ---- halo pieces of this wave: k = wave + 8 s, s = 0..5 (k < 42): pieces 0..20 = hi plane, 21..41 = lo plane, each plane
    // a linear image of 336 pixel rows x 64 B (324 used).  Lane -> pixel row hp = 16 (k % 21) + lane / 4, physical 16-byte slot
    // lane & 3; the slot swizzle goes on the SOURCE (the DMA destination is lane-linear).  hoff = byte offset from the plane base
    // of the chunk; pixels outside the image (and the 12 pad rows) read the planes' zero row N*H*W.  hoff belongs to the tile
    // whose halo is being put in flight: the tile of the MFMAs or, towards its end, the next one.
    unsigned hoff[6];
    const long long zero_row = (long long)a.N * a.H * a.W;
    auto setup_halo = [&](int t_img, int t_ox0, int t_oy0) {
        // (the lane index laundered through an empty asm: the per-lane halo coordinates of the six pieces are tile-invariant, and
        // hipcc otherwise hoists them out of the tile loop -- 30 more live registers across the k-loop, spills at 8 n-tiles)
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            int k = wave + 8 * s;
            if (k >= 2 * HALO_PLANE_PIECES) k -= 8;      // waves 2..7 have no sixth piece: they send their fifth twice (see below)
            const int kp = k >= HALO_PLANE_PIECES ? k - HALO_PLANE_PIECES : k;
            const int hp = 16 * kp + (ln >> 2);
            const int hy = hp / HW_, hx = hp - hy * HW_;
            const int iy = t_oy0 - 1 + hy, ix = t_ox0 - 1 + hx;
            const bool ok = hp < HALO_PIX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const int ls = (ln & 3) ^ swz64(hp);
            const long long row = ok ? ((long long)t_img * a.H + iy) * a.W + ix : zero_row;
            hoff[s] = (unsigned)(row * 64 + ls * 16);
        }
    };
    setup_halo(img, ox0, oy0);
    const long long chunk_bytes = a.in_rows * 64;
    // ---- DMA schedule.  Everything in the k-loop is branch-free and the same for every wave, so that the vmcnt waits are plain
    // immediates: a taken scalar branch costs 20-40 cycles and the first version of this loop had ten per read phase (validity
    // of a piece, end of the k-steps, a decision tree around s_waitcnt: 230 ticks of a 900-tick phase, tools/stamp_conv3p.py).
    //   * weights: SW pieces per wave and k-step; piece idx = wave + 8 s -> plane idx / WN, row group idx % WN (16 rows x 64 B =
    //     one contiguous KiB of the k-step-major planes).  A wave without a piece s (idx >= 2 WN) sends its piece s - 1 again:
    //     same bytes to the same place.  Behind a tile's last k-step come the first k-steps of the workgroup's next tile (or,
    //     after the last tile, the last k-step again into ring slots nobody reads any more).
    //   * halo of the next chunk: piece wave + 8 t in k-step t = 0..5 (tail k-steps 0 and 1: three pieces each); behind a tile's
    //     last chunk comes chunk 0 of the next tile (after the last tile: the current chunk again into the idle buffer).
    // LDS row i of a 16-row group <- weight row 8 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3): rows 4g..4g+3 of the MFMA result are
    // then channels {0, 8, 4, 12}[g] .. + 3 (see the epilogue)
    const int wrow = 8 * ((lane >> 4) & 1) + 4 * (lane >> 5) + ((lane >> 2) & 3);
    const unsigned wlane = (unsigned)(wrow * 64 + (((lane & 3) ^ swz64(lane >> 2)) << 4));
    const long long step_bytes = (long long)a.wrows * 64;
    const unsigned char* wsrc[SW];          // wave-uniform source of piece s at the next k-step to issue
    const unsigned char* wnext[SW];         // the same at k-step 0 of the workgroup's next tile
    int wdst[SW];                           // its byte offset inside a ring slot
    auto weight_base = [&](int s, int t_n0) -> const unsigned char* {
        int idx = wave + 8 * s;
        if (idx >= 2 * WN) idx -= 8;
        const int ic = idx >= 0 ? idx : 0;                       // (WN < 4: waves >= 2 WN have no piece at all and send piece 0)
        const int icc = ic < 2 * WN ? ic : 0;
        const int plane = icc >= WN ? 1 : 0;
        const int j = icc - plane * WN;
        int rg = t_n0 + 16 * j;
        if (rg >= a.wrows) rg = a.wrows - 16;                    // row groups past the packed rows: columns never stored
        return reinterpret_cast<const unsigned char*>(plane ? w_lo : w_hi) + (long long)rg * 64;
    };
#pragma unroll
    for (int s = 0; s < SW; ++s) {
        int idx = wave + 8 * s;
        if (idx >= 2 * WN) idx -= 8;
        const int ic = idx >= 0 ? idx : 0;
        const int icc = ic < 2 * WN ? ic : 0;
        const int plane = icc >= WN ? 1 : 0;
        const int j = icc - plane * WN;
        wsrc[s] = weight_base(s, n0);
        wnext[s] = weight_base(s, nn0);
        wdst[s] = (plane * BN + 16 * j) * 64;
    }

    const int nchunks = nfull + (ktail ? 1 : 0);
    const int nk = 9 * nfull + (ktail ? 3 : 0);

    int wr_off = 0;                       // ring slot (byte offset) the next weight issue goes to
    int kleft = nk - 1;                   // k-steps of the issuer's tile after the one whose weights are issued next
    auto issue_weights = [&]() {          // weights of the next k-step -> next ring slot
        unsigned char* dst = smem + 2 * HALO_BYTES + wr_off;
        const bool more = kleft > 0;
#pragma unroll
        for (int s = 0; s < SW; ++s) {
            dma16(wsrc[s] + wlane, dst + wdst[s]);
            wsrc[s] = more ? wsrc[s] + step_bytes : (has_next ? wnext[s] : wsrc[s]);       // (scalar selects: no branch)
        }
        kleft = more ? kleft - 1 : (has_next ? nk - 1 : 0);
        wr_off = wr_off + WSLOT == NB * WSLOT ? 0 : wr_off + WSLOT;
    };
    const unsigned char* hsrc_hi = reinterpret_cast<const unsigned char*>(in_hi);      // plane bases of the chunk whose halo is issued next
    const unsigned char* hsrc_lo = reinterpret_cast<const unsigned char*>(in_lo);
    int hbuf = 0;                         // halo buffer (byte offset) that chunk goes to
    auto issue_halo = [&](auto sc) {      // halo piece wave + 8 S (waves 2..7, S = 5: piece wave + 32 again)
        constexpr int S = decltype(sc)::value;
        int k = wave + 8 * S;
        if (S == 5 && k >= 2 * HALO_PLANE_PIECES) k -= 8;
        dma16((k >= HALO_PLANE_PIECES ? hsrc_lo : hsrc_hi) + hoff[S], smem + hbuf + k * 1024);
    };
    int chunks_left = nchunks - 1;        // chunks of the issuer's tile after the one whose halo is issued next
    // all six pieces of a chunk are out: on to the tile's next chunk, to chunk 0 of the next tile, or (after the last tile) nowhere
    auto halo_advance = [&]() {
        if (chunks_left > 0) {
            hsrc_hi += chunk_bytes;
            hsrc_lo += chunk_bytes;
            --chunks_left;
        } else if (has_next) {
            hsrc_hi = reinterpret_cast<const unsigned char*>(in_hi);
            hsrc_lo = reinterpret_cast<const unsigned char*>(in_lo);
            setup_halo(nimg, nox0, noy0);
            chunks_left = nchunks - 1;
        }
        hbuf = HALO_BYTES - hbuf;
    };

    f32x4 acc[2][WN], cor[2][WN];

    // tail k-steps: lane group g reads slot 0 of the halo pixel of tap 4t + g (taps 9..11 meet zero weights: tap 8 again)
    int dtail = 0;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int tap = (4 * t + g) < 9 ? 4 * t + g : 8;
        const int ty = tap / 3;
        dtail |= (ty * HW_ + (tap - 3 * ty)) << (8 * t);
    }

    PSTAMP_DECL
    // ---- prologue of the first tile: epilogue constants, halo of chunk 0, weights of k-steps 0 .. LA-1; everything lands before
    // the first read (later tiles: the wait at the tile boundary), so the first LA - 1 k-steps of a tile wait for nothing ----
    dma_planes_consts<BN>(a.bias, a.prelu, a.plane_prelu, a.Cout, n0, cst_base, wave, lane);
    static_for<0, 6>([&](auto sc) { issue_halo(sc); });
    halo_advance();
#pragma unroll
    for (int u = 0; u < LA; ++u) issue_weights();
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();          // the second group runs one phase behind
    TSTAMP(0)

    f16x8 xh[2], xl[2], wh[WN], wl[WN];
    const unsigned wfrag = ring0 + (unsigned)(r * 64 + ((g ^ swz64(r)) << 4));       // swz64(16 j + r) == swz64(r)
    const int prow = 2 * wave * HW_ + r;                                                // halo pixel of (row 2 wave, column r), tap (0,0)
    // Activation-fragment addresses without per-read arithmetic: the fragment of tap offset C sits at pixel p = prow + C, slot
    // g ^ swz64(p), and swz64(p) only depends on bit 2 of p, i.e. on prow and C mod 8: eight per-lane bases XA[C & 7] (current halo
    // buffer; moved to the other buffer once per chunk) + the immediate 64 * C.
    unsigned xa[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) xa[k] = halo0 + (unsigned)(prow * 64 + ((g ^ swz64(prow + k)) << 4));
    int rd_off = 0;                       // ring slot (byte offset) of the k-step being read
    int hcur = 0;                         // halo buffer of the chunk being read
    int seq = 0;                          // tiles done by this workgroup

    // One k-step of one wave.  T = tap (regular chunk) or tail step; FIRST: the tile's first chunk.
    auto kstep = [&](auto tc, auto tailc, auto firstc) {
        constexpr int T = decltype(tc)::value;
        constexpr bool TAIL = decltype(tailc)::value;
        constexpr bool FIRST = decltype(firstc)::value;
        // ---------------- read phase ----------------
        if (PDBG(16)) {
            // (diagnostic: no fragment reads at all -- prices what the partner's LDS traffic costs the MFMA phase)
        } else if constexpr (!TAIL) {
            constexpr int C0 = (T / 3) * HW_ + T % 3, C1 = C0 + HW_;       // tap offsets of the wave's two pixel rows
            lds_read16<64 * C0>(xh[0], xa[C0 & 7]);
            lds_read16<64 * C0 + HALO_LO>(xl[0], xa[C0 & 7]);
            lds_read16<64 * C1>(xh[1], xa[C1 & 7]);
            lds_read16<64 * C1 + HALO_LO>(xl[1], xa[C1 & 7]);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int p = prow + i * HW_ + ((dtail >> (TAIL ? 8 * T : 0)) & 0xff);
                const unsigned addr = halo0 + (unsigned)hcur + (unsigned)(p * 64 + (swz64(p) << 4));
                lds_read16<0>(xh[i], addr);
                lds_read16<HALO_LO>(xl[i], addr);
            }
        }
        const unsigned wa = wfrag + (unsigned)rd_off;
        if (!PDBG(16)) static_for<0, WN>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            lds_read16<j * 1024>(wh[j], wa);
            lds_read16<j * 1024 + BN * 64>(wl[j], wa);
        });
        rd_off = rd_off + WSLOT == NB * WSLOT ? 0 : rd_off + WSLOT;
        PSTAMP(1)
        // DMA: weights LA k-steps ahead first, then the halo piece(s) of the next chunk (k-steps 0..5 of a regular chunk: one;
        // tail k-steps 0 and 1: three -- a tile with a tail has no regular chunk left to carry the next tile's first halo): the
        // wait for a k-step's weights then never covers the halo piece issued in the same phase (vmcnt retires in order, and the
        // halo comes from HBM while the weights come from L2)
        if (!PDBG(2)) issue_weights();
        if constexpr (!TAIL && T < 6) {
            if (!PDBG(1)) issue_halo(std::integral_constant<int, T>{});
        }
        if constexpr (TAIL && T < 2) {
            if (!PDBG(1)) {
                issue_halo(std::integral_constant<int, 3 * T>{});
                issue_halo(std::integral_constant<int, 3 * T + 1>{});
                issue_halo(std::integral_constant<int, 3 * T + 2>{});
            }
        }
        PSTAMP(2)
        // Own pieces of the next k-step landed, own fragment reads complete.  The weights of the next k-step were issued LA - 1
        // read phases ago; behind them went SW weight pieces per phase and the halo pieces of the phases t' = T+1-LA .. T of this
        // chunk (phases of the previous chunk that far back carry none: LA <= 4; counting fewer than are behind is safe).
        if constexpr (FIRST && T < LA - 1) {
            // k-steps 1 .. LA-1 of a tile: their weights were complete before the tile began (prologue / tile boundary).  The first
            // counted wait of a tile therefore comes LA - 1 k-steps after the previous tile's stores went out -- it counts them too
            // (vmcnt is one counter on gfx9), and with the wait one k-step earlier the 101-wide full-resolution layers lose the 4 %
            // the persistent grid gives them.
        } else {
            constexpr int lo = T + 1 - LA > 0 ? T + 1 - LA : 0;
            constexpr int hi = T < 5 ? T : 5;
            constexpr int halos = !TAIL ? (hi >= lo ? hi - lo + 1 : 0) : (lo <= 0 ? 3 : 0) + ((lo <= 1 && T >= 1) ? 3 : 0);
            // (tap 8: the next chunk's halo has to be complete -- its tap-5 piece, which the window formula still counts as "behind" with
            // a lookahead of 4 k-steps, must not stay in flight over the chunk boundary: three k-steps old by then, it never showed)
            if (!PDBG(4)) wait_vm<(!TAIL && T == 8) ? (LA - 1) * SW : (LA - 1) * SW + halos>();
        }
        PSTAMP(3)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(xh[0]), "+v"(xh[1]), "+v"(xl[0]), "+v"(xl[1]));
#pragma unroll
        for (int j = 0; j < WN; ++j) asm volatile("" : "+v"(wh[j]), "+v"(wl[j]));
        __builtin_amdgcn_sched_barrier(0);
        PSTAMP(4)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        PSTAMP(5)
        // ---------------- MFMA phase ----------------
        if (!PDBG(8)) __builtin_amdgcn_s_setprio(1);
        static_for<0, WN>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            cor[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[j], xh[0], cor[0][j], 0, 0, 0);
            cor[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[j], xh[1], cor[1][j], 0, 0, 0);
            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xh[0], acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xh[1], acc[1][j], 0, 0, 0);
            cor[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xl[0], cor[0][j], 0, 0, 0);
            cor[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xl[1], cor[1][j], 0, 0, 0);
        });
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        PSTAMP(6)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        PSTAMP(7)
    };
    // the chunk being read moves to the other halo buffer; so does the chunk whose halo is issued next
    auto next_chunk = [&]() {
        const int d = hcur ? -HALO_BYTES : HALO_BYTES;
#pragma unroll
        for (int k = 0; k < 8; ++k) xa[k] += d;
        hcur += d;
        halo_advance();
    };

    // fused read-out: the lane's fragments of W2 stay in registers for all of the workgroup's tiles (loaded in every tile's epilogue they
    // cost a round trip to L2 per tile with nothing to hide it; the 2- and 4-n-tile instances have 80+ registers to spare)
    constexpr int KK2 = (WN == 2 || WN == 4) ? WN / 2 : 1;
    f16x8 w2h[2][KK2], w2l[2][KK2];
    if constexpr (WN == 2 || WN == 4) {
        if (a.h2_w) {
            const f16x8* wp = reinterpret_cast<const f16x8*>(a.h2_w);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int kk = 0; kk < KK2; ++kk) {
                    w2h[t][kk] = wp[(t * KK2 + kk) * 64 + lane];
                    w2l[t][kk] = wp[((2 + t) * KK2 + kk) * 64 + lane];
                }
        }
    }
    for (;;) {
        const float* cst = cst_base + (seq & 1) * CSTF;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                cor[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                // pinned: hipcc otherwise folds the zeros into the first MFMAs' C operand and may use the (then dead) accumulator
                // registers as temporaries of the first read phase, guarded by s_waitcnt vmcnt(0)
                asm volatile("" : "+v"(acc[i][j]), "+v"(cor[i][j]));
            }
        if (nfull > 0) {
            static_for<0, 9>([&](auto tc) { kstep(tc, std::false_type{}, std::true_type{}); });
            next_chunk();
        }
        for (int c = 1; c < nfull; ++c) {
            static_for<0, 9>([&](auto tc) { kstep(tc, std::false_type{}, std::false_type{}); });
            next_chunk();
        }
        if (ktail) {
            static_for<0, 3>([&](auto tc) { kstep(tc, std::true_type{}, std::false_type{}); });
            next_chunk();
        }
        // ---- tile boundary.  The first group waits for the second one's last MFMA phase; the next tile's epilogue constants go
        // out; everything else in flight -- the next tile's first LA k-steps of weights and its first halo -- has to be complete
        // before that tile's first read, and is waited for HERE, before this tile's stores join the queue (vmcnt counts them).
#if !defined(ATMVFI_STAMP_KSTEP)
        TSTAMP(1)             // the tile's k-loop
#endif
        if (grp == 0) __builtin_amdgcn_s_barrier();
        if (has_next) {
            dma_planes_consts<BN>(a.bias, a.prelu, a.plane_prelu, a.Cout, nn0, cst_base + ((seq + 1) & 1) * CSTF, wave, lane);
            wait_vm<1>();             // all but the constants: the next tile's first LA k-steps of weights and its first halo
        } else {
            wait_vm<0>();
        }
#if !defined(ATMVFI_STAMP_KSTEP)
        TSTAMP(2)             // the first group's wait for the second one's last MFMA phase + the wait for the next tile's first DMA
#endif

    // ---- PReLU as max(v, s v).  For 0 <= s <= 1 that IS v > 0 ? v : s v, bit for bit (s v lies between v and 0, signed zeros included),
    // in 1.5 VALU instructions per value (v_pk_mul_f32 + v_max_f32) instead of 2.5 and two wait states (v_cmp -> vcc -> v_cndmask);
    // a tile whose constants hold a slope outside [0, 1] (or a NaN) takes the select form.  Both activations of a decoder tile are
    // ~45 % of its epilogue's VALU work, and the epilogue is VALU-bound (tools/stamp_conv3p.py).  The check reads the tile's two slope
    // rows from LDS (absent arrays and columns past Cout hold 1).
    constexpr bool MAXFORM = WN < 8;
    bool slopes_in_range = true;
#pragma unroll
    for (int k = 0; MAXFORM && k < (2 * BN + 63) / 64; ++k) {
        const int idx = BN + 64 * k + lane;
        const float sv = cst[idx < 3 * BN ? idx : 3 * BN - 1];
        slopes_in_range = slopes_in_range && sv >= 0.f && sv <= 1.f;
    }
    const bool pmax = __builtin_amdgcn_ballot_w64(!slopes_in_range) == 0ull;
    // (one uniform branch per STAGE -- the fold / bias / activation loop, each plane sink -- chooses the form: a branch around every
    // activation costs more scalar time than the form saves, two copies of the whole epilogue cost 16-60 registers)
    auto prelu4 = [](auto mode, const f32x4 v, const f32x4 sl, const bool on) -> f32x4 {
        constexpr int MODE = decltype(mode)::value;          // 0: no activation, 1: select form, 2: max form, 3: select form if `on`
        if constexpr (MODE == 3) {
            f32x4 o = v;
            if (on) o = (f32x4){v.x > 0.f ? v.x : sl.x * v.x, v.y > 0.f ? v.y : sl.y * v.y, v.z > 0.f ? v.z : sl.z * v.z, v.w > 0.f ? v.w : sl.w * v.w};
            return o;
        } else
        if constexpr (MODE == 2) {
            const f32x4 t = sl * v;
            f32x4 o;
            // (v_max_f32 by hand: fmaxf() makes hipcc canonicalize each operand first -- a second v_max_f32 per value)
            asm("v_max_f32 %0, %1, %2" : "=v"(o.x) : "v"(v.x), "v"(t.x));
            asm("v_max_f32 %0, %1, %2" : "=v"(o.y) : "v"(v.y), "v"(t.y));
            asm("v_max_f32 %0, %1, %2" : "=v"(o.z) : "v"(v.z), "v"(t.z));
            asm("v_max_f32 %0, %1, %2" : "=v"(o.w) : "v"(v.w), "v"(t.w));
            return o;
        } else if constexpr (MODE == 1) {
            return (f32x4){v.x > 0.f ? v.x : sl.x * v.x, v.y > 0.f ? v.y : sl.y * v.y, v.z > 0.f ? v.z : sl.z * v.z, v.w > 0.f ? v.w : sl.w * v.w};
        } else {
            return v;
        }
    };
    typedef std::integral_constant<int, 0> ActNone;
    typedef std::integral_constant<int, 1> ActSelect;
    typedef std::integral_constant<int, 2> ActMax;
    typedef std::integral_constant<int, 3> ActRuntime;
    // ---- epilogue.  Lane (r, g) holds rows 4g..4g+3 of every 16-row n-tile = channels cb(g)..cb(g)+3 with cb = {0, 8, 4, 12}:
    // the weight rows were permuted that way on their way into LDS (wlane), so that lanes g and g + 2 -- the two halves of the
    // wave, which v_permlane32_swap exchanges -- hold the two halves of one 8-channel group.
    const int cb = 8 * (g & 1) + 4 * (g >> 1);
    float* orow[2];
    long long prow_o[2];
    bool live[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int oy = oy0 + 2 * wave + i, ox = ox0 + r;
        live[i] = oy < a.H && ox < a.W;
        prow_o[i] = ((long long)img * a.H + (live[i] ? oy : 0)) * a.W + (live[i] ? ox : 0);
        orow[i] = out_f32 ? out_f32 + prow_o[i] * a.out_ld : nullptr;
    }
    f32x4 vv[2][WN];
    auto fold = [&](auto mode) {
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int cl = 16 * j + cb;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(cst + cl);
            const f32x4 pv = *reinterpret_cast<const f32x4*>(cst + BN + cl);
#pragma unroll
            for (int i = 0; i < 2; ++i) vv[i][j] = prelu4(mode, acc[i][j] + cor[i][j] * LO_UNSCALE + bv, pv, a.prelu != nullptr);
        }
    };
    // (the 8-n-tile instance has no register to spare for copies of a stage -- it spills 32-37 -- and its layers have the longest K: it
    // keeps one copy with the select form under a uniform flag)
    if constexpr (!MAXFORM) fold(ActRuntime{});
    else if (!a.prelu) fold(ActNone{});
    else if (pmax) fold(ActMax{});
    else fold(ActSelect{});
    if constexpr (WN == 2 || WN == 4) {
        if (a.h2_w) {
            constexpr int KK = WN / 2;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x4 a2[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
                f32x4 c2[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) {
                    // the lane's eight k values of this k-step: its four channels of n-tile 2 kk and of n-tile 2 kk + 1
                    const f32x4 u0 = vv[i][2 * kk], u1 = vv[i][2 * kk + 1];
                    f16x2 h0, l0, h1, l1, h2, l2, h3, l3;
                    split_pair((f32x2){u0.x, u0.y}, h0, l0);
                    split_pair((f32x2){u0.z, u0.w}, h1, l1);
                    split_pair((f32x2){u1.x, u1.y}, h2, l2);
                    split_pair((f32x2){u1.z, u1.w}, h3, l3);
                    const f16x8 bh = {h0.x, h0.y, h1.x, h1.y, h2.x, h2.y, h3.x, h3.y};
                    const f16x8 bl = {l0.x, l0.y, l1.x, l1.y, l2.x, l2.y, l3.x, l3.y};
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        c2[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2l[t][kk], bh, c2[t], 0, 0, 0);
                        a2[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2h[t][kk], bh, a2[t], 0, 0, 0);
                        c2[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2h[t][kk], bl, c2[t], 0, 0, 0);
                    }
                }
                // rows 4 g .. 4 g + 3 of row tile t = contributions (tap * 3 + o) = 16 t + 4 g + e of pixel r: planar, so that the 16
                // lanes of a row group write 64 contiguous bytes and the gather pass reads coalesced
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const f32x4 tv = a2[t] + c2[t] * LO_UNSCALE;
                    const int row = 16 * t + 4 * g;
                    float* tp = a.h2_out + (long long)row * a.h2_plane + prow_o[i];
                    if (live[i]) {
                        if (row + 0 < 27) tp[0] = tv.x;
                        if (row + 1 < 27) tp[a.h2_plane] = tv.y;
                        if (row + 2 < 27) tp[2 * a.h2_plane] = tv.z;
                        if (row + 3 < 27) tp[3 * a.h2_plane] = tv.w;
                    }
                }
            }
        }
    }
    if (out_f32) {
        // fp32 NHWC rows: 16 bytes per lane, the four lanes of a pixel cover 64 contiguous bytes.  Only channels >= out_cmin are
        // wanted in fp32 (e.g. the five flow / mask channels of a decoder map whose features go on as planes).
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int co = n0 + 16 * j + cb;
            const int nvalid = a.Cout - co;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f32x4 v = vv[i][j];
                if (live[i] && co + 4 > a.out_cmin) {
                    if (nvalid >= 4) {
                        *reinterpret_cast<f32x4*>(orow[i] + co) = v;
                    } else if (nvalid > 0) {
                        orow[i][co] = v.x;
                        if (nvalid > 1) orow[i][co + 1] = v.y;
                        if (nvalid > 2) orow[i][co + 2] = v.z;
                    }
                }
            }
        }
    }
    // Plane sink, fully coalesced: n-tiles in pairs (j, j+1); one swap per dword gives the lower half-wave the 8 channels
    // 16j + 8g .. of n-tile j and the upper half-wave those of n-tile j+1 (T21 of the programming guide), so ONE 16-byte
    // store per lane and plane writes 16 pixels x 32 channels x 2 bytes = one contiguous KiB of the chunk-major plane.
    // Channels past Cout inside the last group of 8 are zero (zero weight rows, zero bias); groups beyond are not stored.
    auto plane_sink = [&](_Float16* phi, _Float16* plo, long long prows, int pc0, auto mode) {
        constexpr bool pslope = decltype(mode)::value != 0;
        const bool slope_on = a.plane_prelu != nullptr;
        const int climit = (a.Cout + 7) & ~7;
        constexpr int NP = (WN + 1) / 2;
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) {
            const int j0 = 2 * jp, j1 = (2 * jp + 1 < WN) ? 2 * jp + 1 : 2 * jp;
            f32x4 s0 = (f32x4){1.f, 1.f, 1.f, 1.f}, s1 = s0;
            if constexpr (pslope) {     // the sink's own slopes: third row of the tile's constants (no global load between the stores)
                s0 = *reinterpret_cast<const f32x4*>(cst + 2 * BN + 16 * j0 + cb);
                s1 = *reinterpret_cast<const f32x4*>(cst + 2 * BN + 16 * j1 + cb);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x4 u0 = vv[i][j0], u1 = vv[i][j1];
                if constexpr (pslope) {
                    u0 = prelu4(mode, u0, s0, slope_on);
                    u1 = prelu4(mode, u1, s1, slope_on);
                }
                f16x2 h00, l00, h01, l01, h10, l10, h11, l11;
                split_pair((f32x2){u0.x, u0.y}, h00, l00);
                split_pair((f32x2){u0.z, u0.w}, h01, l01);
                split_pair((f32x2){u1.x, u1.y}, h10, l10);
                split_pair((f32x2){u1.z, u1.w}, h11, l11);
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                auto bits = [](f16x2 v) { return __builtin_bit_cast(unsigned, v); };
                // vdst = n-tile j0's dword, src = n-tile j1's: lower lanes end up with [own j0 | upper's j0], upper lanes with
                // [lower's j1 | own j1]
                const auto sh0 = __builtin_amdgcn_permlane32_swap(bits(h00), bits(h10), false, false);
                const auto sh1 = __builtin_amdgcn_permlane32_swap(bits(h01), bits(h11), false, false);
                const auto sl0 = __builtin_amdgcn_permlane32_swap(bits(l00), bits(l10), false, false);
                const auto sl1 = __builtin_amdgcn_permlane32_swap(bits(l01), bits(l11), false, false);
                const u32x4 hv = {sh0[0], sh1[0], sh0[1], sh1[1]};
                const u32x4 lv = {sl0[0], sl1[0], sl0[1], sl1[1]};
                const int jt = g < 2 ? j0 : 2 * jp + 1;                       // the n-tile this lane stores
                const int cs = n0 + 16 * jt + 8 * (g & 1);                      // first of its 8 channels
                if (live[i] && jt < WN && cs < climit) {
                    const int c = pc0 + cs;
                    const long long off = ((long long)(c >> 5) * prows + prow_o[i]) * 32 + (c & 31);
                    *reinterpret_cast<u32x4*>(phi + off) = hv;
                    *reinterpret_cast<u32x4*>(plo + off) = lv;
                }
            }
        }
    };
    if (a.out_hi) {
        if constexpr (!MAXFORM) plane_sink(a.out_hi, a.out_lo, a.plane_rows, a.out_c0, ActRuntime{});
        else if (!a.plane_prelu) plane_sink(a.out_hi, a.out_lo, a.plane_rows, a.out_c0, ActNone{});
        else if (pmax) plane_sink(a.out_hi, a.out_lo, a.plane_rows, a.out_c0, ActMax{});
        else plane_sink(a.out_hi, a.out_lo, a.plane_rows, a.out_c0, ActSelect{});
    }
    if (a.out_hi2) plane_sink(a.out_hi2, a.out_lo2, a.plane_rows2, a.out_c02, ActNone{});
#ifdef ATMVFI_STAMP
        if (a.stamp) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t_end = __builtin_amdgcn_s_memtime();
            st_epi += t_end - st_t0;
            st_t0 = t_end;
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
        if (!has_next) break;
        // on to the next tile: its first halo and its first LA k-steps of weights are in LDS, the DMA streams are already in it
        vb += grid;
        img = nimg; ox0 = nox0; oy0 = noy0; n0 = nn0;
        ++seq;
        has_next = decode(vb + grid, nimg, nox0, noy0, nn0);
#pragma unroll
        for (int s = 0; s < SW; ++s) wnext[s] = weight_base(s, nn0);
        if (grp == 1) __builtin_amdgcn_s_barrier();          // the second group drops one phase behind again
#if !defined(ATMVFI_STAMP_KSTEP)
        TSTAMP(3)             // the next-but-one tile's decode + the second group's extra barrier
#endif
    }
#ifdef ATMVFI_STAMP
    if (a.stamp && lane == 0) {
        unsigned long long* o = a.stamp + ((long long)blockIdx.x * 8 + wave) * 10;
        const unsigned long long nt = (unsigned long long)(seq + 1);
        o[0] = st_[0];                                   // prologue of the first tile
        for (int k = 1; k < 8; ++k) o[k] = st_[k] / nt;  // per tile
        o[8] = st_epi / nt;                              // tile boundary + epilogue, per tile
        o[9] = (unsigned long long)nk;
    }
#endif
}

template <int WN>
int launch_planes(const Conv3PDev& d, int ntiles, hipStream_t s) {
    constexpr int BN = 16 * WN;
    const size_t lds = (size_t)2 * HALO_BYTES + (size_t)ring_slots(WN) * 2 * BN * 64 + 2 * planes_const_floats(BN) * sizeof(float);
    static_assert(2 * HALO_BYTES + ring_slots(WN) * 2 * BN * 64 + 2 * planes_const_floats(BN) * 4 <= 160 * 1024, "LDS budget");
    auto kern = conv3x3_planes_kernel<WN>;
    const hipError_t attr_err = atmvfi::allow_dynamic_lds<conv3x3_planes_kernel<WN>>(lds);
    ATMVFI_REQUIRE(attr_err == hipSuccess, ATMVFI_ELAUNCH, "conv3x3_planes: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    Conv3PDev ds = d;
    ds.nblocks = (ntiles + WN - 1) / WN;
    ds.tiles_y = (d.H + 15) / 16;
    const long long sgroups = ((long long)d.N * d.tiles_x * ds.tiles_y + 7) / 8;
    ds.tchunk = (int)sgroups;
#ifdef ATMVFI_STAMP
    ds.stamp = g_planes_stamp;
    static const int dbg = [] { const char* e = getenv("ATMVFI_P3_DBG"); return e ? atoi(e) : 0; }();
    ds.dbg = dbg;
#else
    ds.stamp = nullptr;
    ds.dbg = 0;
#endif
    ATMVFI_REQUIRE(sgroups * 8 * ds.nblocks < (1LL << 31), ATMVFI_EINVAL, "conv3x3_planes: grid too large");
    // q = (mul_hi(n, m) + n) >> s with s = ceil(log2 d), m = ceil(2^(32 + s) / d) - 2^32: exact for every n < 2^31 and d >= 1
    auto magic = [](int d, unsigned& m, unsigned& sh) {
        sh = 0;
        while ((1ll << sh) < d) ++sh;
        m = (unsigned)((((unsigned long long)1 << (32 + sh)) + (unsigned)d - 1) / (unsigned)d - ((unsigned long long)1 << 32));
    };
    magic(ds.nblocks, ds.dm_nblocks, ds.ds_nblocks);
    magic(d.tiles_x * ds.tiles_y, ds.dm_perimg, ds.ds_perimg);
    magic(8 * d.tiles_x, ds.dm_grp, ds.ds_grp);
    magic((ds.tiles_y & 7) ? (ds.tiles_y & 7) : 8, ds.dm_rows, ds.ds_rows);
    ds.vblocks = d.ksplit > 1 ? (int)((long long)d.N * d.tiles_x * ds.tiles_y * ds.nblocks) : (int)(sgroups * 8 * ds.nblocks);
    // persistent (one workgroup per CU walking its XCD's tiles, DMA streams flowing across tiles) whenever a tile has at least two
    // 32-channel chunks -- the halo stream moves on to the next tile while the last chunk is consumed; else one workgroup per tile
    const int nchunks = (d.cf >> 5) + (d.tail ? 1 : 0);
    if (d.ksplit > 1) {
        // split-K: one workgroup per (tile, K range), no persistence (the launch exists because the tiles alone leave CUs idle)
        hipLaunchKernelGGL(kern, dim3((unsigned)ds.vblocks, (unsigned)d.ksplit), dim3(512), lds, s, ds);
        return atmvfi::check_launch("conv3x3_planes (split-K)");
    }
    const int grid = nchunks >= 2 ? std::min(ds.vblocks, atmvfi::cu_count()) : ds.vblocks;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, s, ds);
    return atmvfi::check_launch("conv3x3_planes");
}

// Split-K, second half: out = epilogue(sum over splits, in split order, of the partial sums + bias) -- the epilogue of the kernel above
// (PReLU, fp32 rows from out_cmin on, one or two plane sinks, the first through its own PReLU; channels from Cout up to the next multiple
// of 8 are written to the planes as zero).  One thread per (pixel, 4 channels); fixed summation order: run-to-run deterministic.
__global__ void conv3_splitk_reduce_kernel(const float* __restrict__ part, long long part_stride, int S, long long rows, int ldp, int Cout,
                                           const float* __restrict__ bias, const float* __restrict__ prelu, float* __restrict__ out, int out_ld,
                                           int out_cmin, _Float16* hi, _Float16* lo, long long prows, int c0, const float* __restrict__ plane_prelu,
                                           _Float16* hi2, _Float16* lo2, long long prows2, int c02) {
    fp16_saturate_on();
    const int groups = ((Cout + 7) & ~7) >> 2;
    const long long total = rows * groups;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long row = idx / groups;
        const int c = (int)(idx - row * groups) * 4;
        const int nvalid = Cout - c;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        // per-channel vectors of the group (bias / slope arrays are 16-byte aligned and c is a multiple of 4; a ragged last group reads
        // element by element -- selects, no indexed local arrays: hipcc turns those into LDS / scratch)
        auto vec4 = [&](const float* a, float dflt) -> f32x4 {
            if (!a || nvalid <= 0) return (f32x4){dflt, dflt, dflt, dflt};
            if (nvalid >= 4) return *reinterpret_cast<const f32x4*>(a + c);
            f32x4 o = (f32x4){a[c], dflt, dflt, dflt};
            if (nvalid > 1) o.y = a[c + 1];
            if (nvalid > 2) o.z = a[c + 2];
            return o;
        };
        if (nvalid > 0) {
            const float* p = part + row * ldp + c;               // ldp is a multiple of 16 floats, c of 4: 16-byte aligned
            for (int sp = 0; sp < S; ++sp) v += *reinterpret_cast<const f32x4*>(p + sp * part_stride);
            v += vec4(bias, 0.f);
            if (prelu) {
                const f32x4 sl = vec4(prelu, 1.f);
                v.x = v.x > 0.f ? v.x : sl.x * v.x;  v.y = v.y > 0.f ? v.y : sl.y * v.y;
                v.z = v.z > 0.f ? v.z : sl.z * v.z;  v.w = v.w > 0.f ? v.w : sl.w * v.w;
            }
            if (nvalid < 4) {                                       // channels past Cout: zero in the planes, never stored in fp32
                v.y = nvalid > 1 ? v.y : 0.f;
                v.z = nvalid > 2 ? v.z : 0.f;
                v.w = 0.f;
            }
        }
        if (out && nvalid > 0 && c + 4 > out_cmin) {
            float* o = out + row * out_ld + c;
            if (nvalid >= 4) *reinterpret_cast<f32x4*>(o) = v;
            else { o[0] = v.x; if (nvalid > 1) o[1] = v.y; if (nvalid > 2) o[2] = v.z; }
        }
        if (hi) {
            f32x4 u = v;
            if (plane_prelu && nvalid > 0) {
                const f32x4 sl = vec4(plane_prelu, 1.f);
                u.x = u.x > 0.f ? u.x : sl.x * u.x;  u.y = u.y > 0.f ? u.y : sl.y * u.y;
                u.z = u.z > 0.f ? u.z : sl.z * u.z;  u.w = u.w > 0.f ? u.w : sl.w * u.w;
            }
            sink_store4(RowSink{nullptr, 0, hi, lo, prows}, row, c0 + c, u);
        }
        if (hi2) sink_store4(RowSink{nullptr, 0, hi2, lo2, prows2}, row, c02 + c, v);
    }
}

// Tile width and K split of a launch.  Width: rounds x (WN + c0), one workgroup per CU (conv3x3_f16x3_row.hip's cost model, row
// schedule).  Split: only when the tiles of the launch leave at least half of the CUs idle and K is long (>= 8 chunks = 72 k-steps):
// as many K ranges as fill the chip, each at least 4 chunks, at most 8 (tools/profile_layers.py at 256 x 256 / 576 x 960: the motion
// MLPs there are 22-72 workgroups walking 200-380 k-steps each: 62 / 161 us per launch on a mostly idle chip).
struct Conv3Plan { int wn, ksplit, cps; };
static Conv3Plan conv3_plan(int N, int H, int W, int Cin, int Cout, int wn, bool may_split) {
    const int ntiles = (Cout + 15) / 16;
    const int tiles_x = (W + TW - 1) / TW;
    const int ncu = atmvfi::cu_count();
    const long long spatial = (long long)N * tiles_x * ((H + 15) / 16);
    int best = wn;
    if (best == 0) {
        float best_cost = 1e30f;
        int best_pad = 1 << 30;
        for (int w = 1; w <= 8; ++w) {
            const int nb = (ntiles + w - 1) / w;
            const float c = (float)((spatial * nb + ncu - 1) / ncu) * ((float)w + 2.0f);
            // ties go to the width with fewer padded n-tiles, then to the wider one (round 4, tools/sweep_conv3p_wn.py: the 576-wide
            // motion-MLP layers at 136 x 240 tie between 8 tiles x 5 blocks, four of the 40 n-tiles padding, and 4 x 9: 0.608 against
            // 0.572 ms and 0.459 against 0.427)
            const int pad = nb * w - ntiles;
            if (c < best_cost || (c == best_cost && pad <= best_pad)) { best_cost = c; best = w; best_pad = pad; }
        }
    }
    Conv3Plan p{best, 1, 0};
    const int t = Cin % 32;
    const int nfull = (t >= 1 && t <= 8) ? (Cin - t) / 32 : (Cin + 31) / 32;
    if (may_split && nfull >= 8) {
        // A split launch is one round by construction; its time is the k-steps of one K range times the k-step time of the width
        // (~ WN + 2, the same model; one unit = 0.75-0.9 us per chunk of 9 k-steps on an under-filled chip, tools/sweep_conv3p_splitk.py)
        // plus the second kernel (~7 us = 9 units).  Splits: as many as fill the chip, each at least 4 chunks, at most 8.  Taken only
        // when the model says at least 10 % faster than the unsplit launch at its own best width.
        const int nb0 = (ntiles + best - 1) / best;
        const long long unsplit = ((spatial * nb0 + ncu - 1) / ncu) * (long long)nfull * (best + 2);
        int bw = 0, bs = 1;
        long long bcost = 1ll << 60;
        for (int w = (wn ? wn : 1); w <= (wn ? wn : 8); ++w) {
            const long long tiles = spatial * ((ntiles + w - 1) / w);
            const int S = (int)std::min<long long>(std::min<long long>(ncu / std::max<long long>(tiles, 1), nfull / 4), 8);
            if (S < 2) continue;
            const long long c = (long long)((nfull + S - 1) / S) * (w + 2) + 9;
            if (c <= bcost) { bcost = c; bw = w; bs = S; }
        }
        if (bs >= 2 && 10 * bcost < 9 * unsplit) {
            p.wn = bw;
            p.ksplit = bs;
            p.cps = nfull / bs;
        }
    }
    return p;
}

}  // namespace

extern "C" int64_t atmvfi_conv3x3_planes_workspace_floats(int N, int H, int W, int Cin, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
    const Conv3Plan p = conv3_plan(N, H, W, Cin, Cout, 0, true);
    return p.ksplit > 1 ? (int64_t)p.ksplit * N * H * W * atmvfi::round_up(Cout, 16) : 0;
}

static int conv3x3_planes_impl(const void* in_hi, const void* in_lo, int64_t in_rows, int N, int H, int W, int Cin, const void* w_hi,
                               const void* w_lo, int Cout, float* out, int out_ld, const float* bias, const float* prelu, void* out_hi,
                               void* out_lo, int64_t plane_rows, int out_c0, const float* plane_prelu, void* out_hi2, void* out_lo2,
                               int64_t plane_rows2, int out_c02, int out_cmin, int wn, float* workspace, int64_t workspace_floats,
                               const void* h2_w, float* h2_out, int64_t h2_plane, void* stream);

extern "C" int atmvfi_conv3x3_planes3(const void* in_hi, const void* in_lo, int64_t in_rows, int N, int H, int W, int Cin, const void* w_hi,
                                       const void* w_lo, int Cout, float* out, int out_ld, const float* bias, const float* prelu,
                                       void* out_hi, void* out_lo, int64_t plane_rows, int out_c0, const float* plane_prelu, void* out_hi2,
                                       void* out_lo2, int64_t plane_rows2, int out_c02, int out_cmin, int wn, float* workspace,
                                       int64_t workspace_floats, void* stream) {
    ATMVFI_REQUIRE(in_hi && in_lo && w_hi && w_lo && (out || out_hi), ATMVFI_EINVAL, "conv3x3_planes: null pointer");
    return conv3x3_planes_impl(in_hi, in_lo, in_rows, N, H, W, Cin, w_hi, w_lo, Cout, out, out_ld, bias, prelu, out_hi, out_lo, plane_rows, out_c0,
                               plane_prelu, out_hi2, out_lo2, plane_rows2, out_c02, out_cmin, wn, workspace, workspace_floats, nullptr, nullptr, 0,
                               stream);
}

static int conv3x3_planes_impl(const void* in_hi, const void* in_lo, int64_t in_rows, int N, int H, int W, int Cin, const void* w_hi,
                               const void* w_lo, int Cout, float* out, int out_ld, const float* bias, const float* prelu, void* out_hi,
                               void* out_lo, int64_t plane_rows, int out_c0, const float* plane_prelu, void* out_hi2, void* out_lo2,
                               int64_t plane_rows2, int out_c02, int out_cmin, int wn, float* workspace, int64_t workspace_floats,
                               const void* h2_w, float* h2_out, int64_t h2_plane, void* stream) {
    ATMVFI_REQUIRE((out_hi2 == nullptr) == (out_lo2 == nullptr), ATMVFI_EINVAL, "conv3x3_planes: the second plane sink needs both planes");
    if (out_hi2)
        ATMVFI_REQUIRE(out_hi && plane_rows2 >= (int64_t)N * H * W && out_c02 >= 0 && out_c02 % 8 == 0 && atmvfi::aligned16(out_hi2) &&
                           atmvfi::aligned16(out_lo2), ATMVFI_EINVAL,
                       "conv3x3_planes: the second plane sink needs the first one, plane_rows2 >= N*H*W, a channel offset that is a multiple "
                       "of 8 and 16-byte aligned planes");
    ATMVFI_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, ATMVFI_EINVAL, "conv3x3_planes: bad shape");
    ATMVFI_REQUIRE(in_rows > (int64_t)N * H * W && in_rows * 64 < (1ll << 32), ATMVFI_EINVAL,
                   "conv3x3_planes: in_rows %lld must exceed N*H*W (the zero row) and in_rows * 64 must fit 32 bits", (long long)in_rows);
    ATMVFI_REQUIRE(atmvfi::aligned16(in_hi) && atmvfi::aligned16(in_lo) && atmvfi::aligned16(w_hi) && atmvfi::aligned16(w_lo) &&
                       (!bias || atmvfi::aligned16(bias)) && (!prelu || atmvfi::aligned16(prelu)),
                   ATMVFI_EALIGN, "conv3x3_planes: pointers (incl. bias/prelu) must be 16-byte aligned");
    if (out)
        ATMVFI_REQUIRE(atmvfi::aligned16(out) && out_ld % 4 == 0 && out_cmin >= 0 && out_cmin % 4 == 0 &&
                           out_ld >= atmvfi::round_up(Cout, 4) - out_cmin,
                       ATMVFI_EALIGN, "conv3x3_planes: fp32 output must be 16-byte aligned with ld %% 4 == 0 covering the stored channels");
    ATMVFI_REQUIRE((out_hi == nullptr) == (out_lo == nullptr), ATMVFI_EINVAL, "conv3x3_planes: plane sink needs both planes");
    if (out_hi) {
        ATMVFI_REQUIRE(plane_rows >= (int64_t)N * H * W && out_c0 >= 0 && out_c0 % 8 == 0, ATMVFI_EINVAL,
                       "conv3x3_planes: plane sink needs plane_rows >= N*H*W and a channel offset that is a multiple of 8");
        ATMVFI_REQUIRE(atmvfi::aligned16(out_hi) && atmvfi::aligned16(out_lo) && (!plane_prelu || atmvfi::aligned16(plane_prelu)),
                       ATMVFI_EALIGN, "conv3x3_planes: plane sink pointers must be 16-byte aligned");
    }
    ATMVFI_REQUIRE(wn >= 0 && wn <= 8, ATMVFI_EINVAL, "conv3x3_planes: wn 0 (auto) or 1..8");
    ATMVFI_REQUIRE(out_cmin >= 0 && out_cmin % 4 == 0, ATMVFI_EINVAL, "conv3x3_planes: out_cmin must be a non-negative multiple of 4");
    Conv3PDev d;
    d.in_hi = (const _Float16*)in_hi; d.in_lo = (const _Float16*)in_lo; d.in_rows = in_rows;
    d.N = N; d.H = H; d.W = W; d.Cin = Cin;
    d.w_hi = (const _Float16*)w_hi; d.w_lo = (const _Float16*)w_lo;
    d.wrows = atmvfi::round_up(Cout, 16);
    const int t = Cin % 32;
    if (t >= 1 && t <= 8) { d.cf = Cin - t; d.tail = t; } else { d.cf = atmvfi::round_up(Cin, 32); d.tail = 0; }
    d.Cout = Cout; d.out = out; d.out_ld = out_ld; d.bias = bias; d.prelu = prelu;
    d.out_hi = (_Float16*)out_hi; d.out_lo = (_Float16*)out_lo; d.plane_rows = plane_rows; d.out_c0 = out_c0; d.plane_prelu = plane_prelu;
    d.out_hi2 = (_Float16*)out_hi2; d.out_lo2 = (_Float16*)out_lo2; d.plane_rows2 = plane_rows2; d.out_c02 = out_c02;
    d.out_cmin = out_cmin;
    d.tiles_x = (W + TW - 1) / TW;
    d.tiles_y = 0; d.nblocks = 0; d.tchunk = 0;
    d.h2_w = (const _Float16*)h2_w; d.h2_out = h2_out; d.h2_plane = h2_plane;
    const int ntiles = (Cout + 15) / 16;
    ATMVFI_REQUIRE(!workspace || atmvfi::aligned16(workspace), ATMVFI_EALIGN, "conv3x3_planes: the split-K workspace must be 16-byte aligned");
    Conv3Plan plan = conv3_plan(N, H, W, Cin, Cout, wn, workspace != nullptr);
    const int ldp = atmvfi::round_up(Cout, 16);
    const long long rows = (long long)N * H * W;
    if (plan.ksplit > 1 && (long long)plan.ksplit * rows * ldp > workspace_floats) plan = conv3_plan(N, H, W, Cin, Cout, wn, false);
    d.ksplit = plan.ksplit; d.cps = plan.cps; d.part_stride = rows * ldp;
    const int best = plan.wn;
    Conv3PDev full = d;
    if (plan.ksplit > 1) {
        // first half: raw partial sums of every K range into the workspace (no bias, no activation, no sinks)
        d.out = workspace; d.out_ld = ldp; d.out_cmin = 0; d.bias = nullptr; d.prelu = nullptr;
        d.out_hi = d.out_lo = d.out_hi2 = d.out_lo2 = nullptr; d.plane_prelu = nullptr;
    }
    hipStream_t s = (hipStream_t)stream;
    int rc;
    switch (best) {
        case 1: rc = launch_planes<1>(d, ntiles, s); break;
        case 2: rc = launch_planes<2>(d, ntiles, s); break;
        case 3: rc = launch_planes<3>(d, ntiles, s); break;
        case 4: rc = launch_planes<4>(d, ntiles, s); break;
        case 5: rc = launch_planes<5>(d, ntiles, s); break;
        case 6: rc = launch_planes<6>(d, ntiles, s); break;
        case 7: rc = launch_planes<7>(d, ntiles, s); break;
        default: rc = launch_planes<8>(d, ntiles, s); break;
    }
    if (rc != ATMVFI_OK || plan.ksplit <= 1) return rc;
    const long long groups = rows * (((Cout + 7) & ~7) >> 2);
    const unsigned blocks = (unsigned)std::min<long long>((groups + 255) / 256, 4096);
    hipLaunchKernelGGL(conv3_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, workspace, d.part_stride, plan.ksplit, rows, ldp, Cout,
                       full.bias, full.prelu, full.out, full.out_ld, full.out_cmin, full.out_hi, full.out_lo, (long long)full.plane_rows,
                       full.out_c0, full.plane_prelu, full.out_hi2, full.out_lo2, (long long)full.plane_rows2, full.out_c02);
    return atmvfi::check_launch("conv3x3_planes (split-K reduce)");
}

extern "C" int atmvfi_conv3x3_planes_readout(const void* in_hi, const void* in_lo, int64_t in_rows, int N, int H, int W, int Cin,
                                              const void* w_hi, const void* w_lo, int Cout, const float* bias, const float* prelu,
                                              const void* w2, float* contrib, int64_t contrib_plane, void* stream) {
    ATMVFI_REQUIRE(in_hi && in_lo && w_hi && w_lo && w2 && contrib, ATMVFI_EINVAL, "conv3x3_planes_readout: null pointer");
    ATMVFI_REQUIRE(Cout == 32 || Cout == 64, ATMVFI_EINVAL, "conv3x3_planes_readout: Cout must be 32 or 64 (one column block of 2 or 4 n-tiles), got %d", Cout);
    ATMVFI_REQUIRE(contrib_plane >= (int64_t)N * H * W && atmvfi::aligned16(w2), ATMVFI_EINVAL,
                   "conv3x3_planes_readout: contrib_plane must cover N*H*W pixels and w2 must be 16-byte aligned");
    return conv3x3_planes_impl(in_hi, in_lo, in_rows, N, H, W, Cin, w_hi, w_lo, Cout, nullptr, 0, bias, prelu, nullptr, nullptr, 0, 0, nullptr,
                               nullptr, nullptr, 0, 0, 0, Cout / 16, nullptr, 0, w2, contrib, contrib_plane, stream);
}

extern "C" int atmvfi_conv3x3_planes2(const void* in_hi, const void* in_lo, int64_t in_rows, int N, int H, int W, int Cin, const void* w_hi,
                                       const void* w_lo, int Cout, float* out, int out_ld, const float* bias, const float* prelu,
                                       void* out_hi, void* out_lo, int64_t plane_rows, int out_c0, const float* plane_prelu, void* out_hi2,
                                       void* out_lo2, int64_t plane_rows2, int out_c02, int out_cmin, int wn, void* stream) {
    return atmvfi_conv3x3_planes3(in_hi, in_lo, in_rows, N, H, W, Cin, w_hi, w_lo, Cout, out, out_ld, bias, prelu, out_hi, out_lo, plane_rows,
                                  out_c0, plane_prelu, out_hi2, out_lo2, plane_rows2, out_c02, out_cmin, wn, nullptr, 0, stream);
}

extern "C" int atmvfi_conv3x3_planes(const void* in_hi, const void* in_lo, int64_t in_rows, int N, int H, int W, int Cin, const void* w_hi,
                                      const void* w_lo, int Cout, float* out, int out_ld, const float* bias, const float* prelu,
                                      void* out_hi, void* out_lo, int64_t plane_rows, int out_c0, const float* plane_prelu, int out_cmin,
                                      int wn, void* stream) {
    return atmvfi_conv3x3_planes2(in_hi, in_lo, in_rows, N, H, W, Cin, w_hi, w_lo, Cout, out, out_ld, bias, prelu, out_hi, out_lo, plane_rows,
                                  out_c0, plane_prelu, nullptr, nullptr, 0, 0, out_cmin, wn, stream);
}

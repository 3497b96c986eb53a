// 3x3 / stride 1 / pad 1 split-precision convolution, "row-stage" schedule (gfx950).
//
// Same tile, arithmetic and LDS images as conv3x3_f16x3.hip (16x16 output pixels x 16*WN channels,
// 512 threads, x = hi + lo'/1024 in fp16, three v_mfma_f32_16x16x32_f16 per product into two fp32
// accumulators), but ONE STAGE = ONE KERNEL ROW: the weights of the three taps (ky, 0..2) of a
// 32-channel chunk sit in one LDS buffer and are consumed between two barriers, so a chunk costs
// 3 stage barriers + 1 halo barrier instead of 9, and every barrier interval carries 3x the MFMA
// work (measured on the one-tap schedule: ~3 000 cycles per stage against 1 344 of MFMA issue; the
// LDS-read, MFMA and LDS-write phases of the 8 lock-stepped waves do not overlap, so the fixed
// part is what has to be amortised).  The halo is single-buffered to make room for the 3-tap
// weight buffers: the next chunk's halo is requested at the start of the chunk's last stage,
// converted and written after that stage's barrier, and published by one extra barrier.
// Weight fragments are read one n-tile ahead of the MFMAs that consume them.
#include "conv3_common.h"

#include <stdlib.h>

#ifdef ATMVFI_STAMP
// Diagnostic build only (`make stamp`, tools/stamp_conv.py): per-wave cycle sums of the phases of a stage.
static unsigned long long* g_stamp_buf = nullptr;
extern "C" void atmvfi_debug_set_stamp_buffer(void* p) { g_stamp_buf = (unsigned long long*)p; }
#define STAMP_DECL unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
#define STAMP(k) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); st_[k] += t1_ - st_t0; st_t0 = t1_; __builtin_amdgcn_sched_barrier(0); }
#else
#define STAMP_DECL
#define STAMP(k)
#endif

// Diagnostic build only (`make ablate`, never the product): ATMVFI_LEGACY_ORDER bits switch pieces of the kernel off (results are
// then wrong) to price them: 2 weight DMA after the prologue, 4 halo reloads, 8 halo convert + write, 16 epilogue stores, 32 the MFMAs
// of the pipelined loop (3+ n-tiles).
#ifdef ATMVFI_ABLATE
#define ABL(bit) ((a.legacy_order & (bit)) != 0)
#else
#define ABL(bit) false
#endif

namespace {

// NWV wavefronts per workgroup (output tile 16 x 2*NWV pixels), TAPS k-steps per stage:
//   <WN, 8, 3>  "row":  512 threads, 16x16 tile, one kernel row per stage, one workgroup per CU (LDS up to 140 KiB)
//   <WN, 4, 1>  "half": 256 threads, 16x8 tile, one tap per stage, 55 KiB of LDS: two workgroups per CU, so one's DMA issue,
//               barrier wait and halo conversion run under the other's MFMAs (at twice the weight traffic per pixel)
template <int WN, int NWV, int TAPS>
__global__ __launch_bounds__(64 * NWV, NWV == 4 ? 2 : 1) void conv3x3_f16x3_row_kernel(const Conv3Dev a) {
    fp16_saturate_on();
    constexpr int BN = 16 * WN;
    constexpr int NT = 64 * NWV;
    constexpr int TH = 2 * NWV, NPIX = HW_ * (TH + 2), HALO_TASKS = NPIX * 8, HALO_TPT = (HALO_TASKS + NT - 1) / NT;
    constexpr int SPC = 9 / TAPS;                       // stages per regular chunk
    extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
    _Float16* halo_hi = smem;                           // [NPIX][32]
    _Float16* halo_lo = halo_hi + NPIX * 32;
    _Float16* b_hi = halo_lo + NPIX * 32;               // [2][TAPS][BN][32]
    _Float16* b_lo = b_hi + 2 * TAPS * BN * 32;
    float* cst = reinterpret_cast<float*>(b_lo + 2 * TAPS * BN * 32);  // bias / PReLU slopes of this column block (common.h)

    const unsigned halo_base = lds_offset(halo_hi), w_base = lds_offset(b_hi);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 15;
    const int g = lane >> 4;

    // XCD-aware order.  Blocks b and b+8 share an XCD and its 4 MiB L2, so (1) the column blocks that re-read one halo tile are
    // dealt to one XCD back to back, and (2) each XCD takes one contiguous eighth of the tiles in an order that keeps halo
    // neighbours close: groups of 8 tile rows walked column by column (vertical neighbours are consecutive, the next column is
    // 8 tiles later), so the 2-of-18 halo rows / columns shared by adjacent tiles are fetched from HBM once.
    const int slot = blockIdx.x >> 3;
    const int sgrp = slot / a.nblocks;
    const int nblk = slot - sgrp * a.nblocks;
    int L = (blockIdx.x & 7) * a.tchunk + sgrp;                  // position in the tile order
    const int per_img = a.tiles_x * a.tiles_y;
    if (a.legacy_order & 1) L = sgrp * 8 + (blockIdx.x & 7);
    if (L >= a.N * per_img) return;
    const int img = L / per_img;
    L -= img * per_img;
    const int grp = L / (8 * a.tiles_x);
    const int rem = L - grp * 8 * a.tiles_x;
    const int rows_here = (a.tiles_y - 8 * grp) < 8 ? a.tiles_y - 8 * grp : 8;
    int txb = rem / rows_here;
    int tyb = 8 * grp + (rem - txb * rows_here);
    if (a.legacy_order & 1) { txb = L % a.tiles_x; tyb = L / a.tiles_x; }
    const int ox0 = txb * TW, oy0 = tyb * TH;      // TH = 2 * NWV rows per tile
    const int n0 = nblk * BN;

    // ---- halo tasks: T = tid + NT*k -> (halo pixel T >> 3, 4-channel group T & 7).  Eight lanes cover the 128 contiguous bytes of
    // one pixel's 32-channel chunk, so a load instruction touches 8 full cache lines (with 4 lanes x 8 channels per pixel and two
    // loads per task it touched 16 half lines each, which costs about twice as much to issue: tools/probes/dma_probe.hip) ----
    const float* hsrc[HALO_TPT];
    int hdst[HALO_TPT], hq[HALO_TPT];
    bool hok[HALO_TPT], hact[HALO_TPT];
#pragma unroll
    for (int k = 0; k < HALO_TPT; ++k) {
        const int T = tid + NT * k;
        hact[k] = T < HALO_TASKS;
        const int hp = hact[k] ? (T >> 3) : 0;
        const int q = T & 7;
        const int hy = hp / HW_, hx = hp - hy * HW_;
        const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
        hok[k] = hact[k] && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        hsrc[k] = a.in + (((long long)img * a.H + (hok[k] ? iy : 0)) * a.W + (hok[k] ? ix : 0)) * a.in_ld + q * 4;
        hdst[k] = hp * 32 + (((q >> 1) ^ swz64(hp)) << 3) + (q & 1) * 4;
        hq[k] = q;
    }
    // ---- weights by LDS-DMA: the fp16 planes go global -> LDS without touching registers.  The planes are k-step major
    // ([k-step][row][32 halves], conv3x3_f16x3.hip), so one wave-instruction = 16 rows x 64 B = one contiguous KiB: wave-piece
    // q = wave + NWV*k -> plane = q / (TAPS*WN), tap-in-stage t = (q / WN) % TAPS, row group q % WN; lane = (row in group,
    // physical slot), the swizzle goes on the SOURCE slot. ----
    constexpr int NWPIECE = 2 * TAPS * WN;                // wave-pieces per stage
    constexpr int NWP = (NWPIECE + NWV - 1) / NWV;        // per wave (the last one may be inactive: wave-uniform)
    const _Float16* wsrc[NWP];
    int wdst[NWP];
    const long long step_halves = (long long)a.wrows * 32;                    // one k-step of one plane
#pragma unroll
    for (int k = 0; k < NWP; ++k) {
        const int qq = wave + NWV * k;
        const int qc = qq < NWPIECE ? qq : 0;
        const int plane = qc / (TAPS * WN);
        const int t = (qc / WN) % TAPS;
        int rg = n0 + 16 * (qc % WN);                                         // first row of the group
        if (rg >= a.wrows) rg = a.wrows - 16;                                 // groups past the packed rows: columns never stored
        const int row = lane >> 2;
        const int ls = (lane & 3) ^ swz64(row);                               // swz64(16j + row) == swz64(row)
        wsrc[k] = (plane ? a.w_lo : a.w_hi) + t * step_halves + (long long)(rg + row) * 32 + ls * 8;
        wdst[k] = plane * (2 * TAPS * BN * 32) + (t * BN + 16 * (qc % WN)) * 32;   // halves from b_hi (b_lo = b_hi + 2*TAPS*BN*32)
    }

    f32x4 acc[2][WN], cor[2][WN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            cor[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }

    // stages: (chunk, kernel row) = three taps x 32 channels each; then ONE stage for the tap-packed channel tail
    // (three k-steps of 4 taps x 8 channels, conv3x3_f16x3.hip) instead of three more
    const int nfull = a.cf >> 5;
    const int nchunks = nfull + (a.tail ? 1 : 0);
    const int nstages = nfull * SPC + (a.tail ? 3 / TAPS : 0);

    f32x4 hr[HALO_TPT];
    int hnv[HALO_TPT];

    auto halo_load = [&](int k, int chunk) {
        // Unconditional loads from a clamped, always-valid address, zeroed afterwards by selects: a
        // predicated load makes hipcc branch around it and wait vmcnt(0) on the spot, which serialises
        // the whole prefetch (seen in the ISA; cdna_hip_programming.md "Three .s-level traps" (c)).
        const int c = chunk * 32 + hq[k] * 4;
        const bool ok = hok[k] && c < a.Cin;
        hnv[k] = ok ? a.Cin - c : 0;                             // valid channels in this group of 4; masking happens at store time
        const float* p = ok ? hsrc[k] + chunk * 32 : a.in;       // masked lanes read the tensor base
        hr[k] = *reinterpret_cast<const f32x4*>(p);
    };
    auto halo_store = [&](int k) {
        if (hact[k]) {
            f32x4 v = hr[k];
            const int nv = hnv[k];
            v.x = nv > 0 ? v.x : 0.f;          // (a wave-uniform branch around the selects makes hipcc wait vmcnt(0) before the
            v.y = nv > 1 ? v.y : 0.f;          //  first task instead of converting each task as its load arrives: +55 % on the
            v.z = nv > 2 ? v.z : 0.f;          //  24-channel full-resolution layer)
            v.w = nv > 3 ? v.w : 0.f;
            f16x2 h0, l0, h1, l1;
            split_pair((f32x2){v.x, v.y}, h0, l0);
            split_pair((f32x2){v.z, v.w}, h1, l1);
            const f16x4 hi = {h0.x, h0.y, h1.x, h1.y}, lo = {l0.x, l0.y, l1.x, l1.y};
            *reinterpret_cast<f16x4*>(halo_hi + hdst[k]) = hi;
            *reinterpret_cast<f16x4*>(halo_lo + hdst[k]) = lo;
        }
    };
    // stage -> (chunk, first k-step q0): regular chunks have 9 k-steps (tap q), the tail chunk 3; k-step q of chunk c is the
    // (9c + q)-th block of the k-step-major weight planes
    auto stage_pos = [&](int stage, int& chunk, int& q0) {
        const int full = nfull * SPC;
        chunk = stage < full ? stage / SPC : nfull;
        q0 = (stage < full ? stage - chunk * SPC : stage - full) * TAPS;
    };
    auto w_dma = [&](int stage, int buf) {
        int chunk, q0;
        stage_pos(stage, chunk, q0);
        const long long koff = (long long)(9 * chunk + q0) * step_halves;
#pragma unroll
        for (int k = 0; k < NWP; ++k)
            if (wave + NWV * k < NWPIECE)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[k] + koff),
                                                 (__attribute__((address_space(3))) void*)(b_hi + buf * TAPS * BN * 32 + wdst[k]), 16, 0, 0);
    };

    int dtail = 0;                 // halo offsets of taps 4t+g, t = 0..2, one byte each (taps 9..11 carry zero weights: any
#pragma unroll                     // finite data will do, tap 8 again); packed: the WN = 8 instance has no register to spare
    for (int t = 0; t < 3; ++t) {
        const int tap = (4 * t + g) < 9 ? 4 * t + g : 8;
        const int ty = tap / 3;
        dtail |= (ty * HW_ + (tap - 3 * ty)) << (8 * t);
    }

    STAMP_DECL
    // ---- prologue ----
    dma_epilogue_consts<BN>(a.bias, a.prelu, n0, cst, wave, lane, [&](int col) { return col < a.Cout ? col : -1; });
#pragma unroll
    for (int k = 0; k < HALO_TPT; ++k) halo_load(k, 0);
    w_dma(0, 0);
#pragma unroll
    for (int k = 0; k < HALO_TPT; ++k) halo_store(k);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // own DMA pieces have landed; the barrier publishes everyone's
    __syncthreads();
    STAMP(0)

    for (int s = 0; s < nstages; ++s) {
        int chunk, q0;
        stage_pos(s, chunk, q0);
        const int wb = s & 1;
        const bool tail_stage = chunk >= nfull;
        int dt = dtail;
        asm volatile("" : "+v"(dt));       // opaque: keeps hipcc from hoisting the six tail-stage offsets out of the loop (spills at WN = 8)
        const bool more_w = (s + 1) < nstages;
        const bool next_halo = !tail_stage && (q0 + TAPS == 9) && (chunk + 1 < nchunks);      // last stage of a regular chunk
        STAMP(7)
        if (more_w && !ABL(2)) w_dma(s + 1, wb ^ 1);       // that buffer was last read in stage s-1: everyone has passed its barrier
        if (next_halo && !ABL(4)) {
#pragma unroll
            for (int k = 0; k < HALO_TPT; ++k) halo_load(k, chunk + 1);
        }
        STAMP(1)
        // One stage = TAPS k-steps x WN n-tiles = NG groups of six MFMAs, run as ONE software pipeline: the weight fragments of
        // group n+2 (3-slot ring) and the activation fragments of the next k-step (second register set) are requested before
        // the MFMAs of group n, and the wait in front of a group leaves the younger reads in flight (lgkmcnt(2..8), FragPipe).
        // hipcc cannot be brought to do this: whatever the source order, it puts s_waitcnt lgkmcnt(0) in front of the first MFMA
        // that needs a fragment, so every k-step drained the LDS queue ~3 times with all 8 waves asking at once, and the kernel
        // ran at LDS time PLUS MFMA time (ablation build: 0.75 ms of a 1.08 ms layer with the MFMAs removed).  The reads and
        // waits are therefore inline asm, invisible to the compiler's counter model; every read is waited for inside the stage.
        if constexpr (WN >= 3) {
            constexpr bool DX = !(WN == 8 && NWV == 8);        // two activation-fragment sets wherever the registers allow
            using P = FragPipe<WN, TAPS, DX>;
            f16x8 xh[DX ? 2 : 1][2], xl[DX ? 2 : 1][2], wh[3], wl[3];
            auto load_x = [&](int t, int set) {
                const int q = q0 + t;                        // k-step: tap q of a regular chunk, step q of the tail
    #pragma unroll
                for (int i = 0; i < 2; ++i) {
                    // regular stage: tap (ky, t), lane group g reads channel slot g.  Tail stage: k = (tap 4t+g, tail channels
                    // 0..7), lane group g reads slot 0 of ITS tap's pixel (selects on a uniform flag: no branch in this loop)
                    const int p = (2 * wave + i) * HW_ + r + (tail_stage ? ((dt >> (8 * q)) & 0xff) : (q / 3) * HW_ + q % 3);
                    const int sl = tail_stage ? 0 : g;
                    const unsigned addr = halo_base + 2u * (unsigned)(p * 32 + ((sl ^ swz64(p)) << 3));
                    lds_read16<0>(xh[set][i], addr);
                    lds_read16<NPIX * 32 * 2>(xl[set][i], addr);
                }
            };
            const unsigned waddr = w_base + 2u * (unsigned)(wb * TAPS * BN * 32 + r * 32 + ((g ^ swz64(r)) << 3));   // swz64(16j + r) == swz64(r)
            const unsigned waddr_lo = waddr + 2u * (2 * TAPS * BN * 32);      // the lo planes (the 16-bit offset field does not reach them)
            load_x(0, 0);
            static_for<0, (P::NG < 2 ? P::NG : 2)>([&](auto nc) {
                constexpr int n = decltype(nc)::value;
                lds_read16<P::w_off(n, BN)>(wh[n % 3], waddr);
                lds_read16<P::w_off(n, BN)>(wl[n % 3], waddr_lo);
            });
            static_for<0, P::NG>([&](auto nc) {
                constexpr int n = decltype(nc)::value;
                constexpr int t = n / WN, j = n % WN;
                if constexpr (n + 2 < P::NG) {
                    lds_read16<P::w_off(n + 2, BN)>(wh[(n + 2) % 3], waddr);
                    lds_read16<P::w_off(n + 2, BN)>(wl[(n + 2) % 3], waddr_lo);
                }
                constexpr int xs = DX ? (t & 1) : 0;
                if constexpr (DX && j == P::JX && t + 1 < TAPS) load_x(t + 1, (t + 1) & 1);
                if constexpr (j == 0) lds_wait4<P::wait(n)>(xh[xs][0], xh[xs][1], xl[xs][0], xl[xs][1]);
                lds_wait2<P::wait(n)>(wh[n % 3], wl[n % 3]);
                const f16x8 ch = wh[n % 3], cl = wl[n % 3];
                const f16x8 h0 = xh[xs][0], h1 = xh[xs][1], l0 = xl[xs][0], l1 = xl[xs][1];
                // dependent MFMAs (same accumulator) are kept 4 issues apart
                if (!ABL(32)) {
                    cor[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cl, h0, cor[0][j], 0, 0, 0);
                    cor[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cl, h1, cor[1][j], 0, 0, 0);
                    acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch, h0, acc[0][j], 0, 0, 0);
                    acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch, h1, acc[1][j], 0, 0, 0);
                    cor[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch, l0, cor[0][j], 0, 0, 0);
                    cor[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch, l1, cor[1][j], 0, 0, 0);
                } else {
                    acc[0][j] += (f32x4){(float)ch[0], (float)cl[0], (float)h0[0], (float)l1[0]};      // keeps the reads alive
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!DX && j == P::JX && t + 1 < TAPS) {
                    load_x(t + 1, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        } else {
            // narrow tiles (1-2 n-tiles: 3- to 48-channel layers, bound by HBM and the halo staging, not by the MFMA loop): the
            // compiler-scheduled form, which measured 35 % faster there than the pinned pipeline
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                const int q = q0 + t;
                f16x8 xh[2], xl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int p = (2 * wave + i) * HW_ + r + (tail_stage ? ((dt >> (8 * q)) & 0xff) : (q / 3) * HW_ + q % 3);
                    const int sl = tail_stage ? 0 : g;
                    const int off = p * 32 + ((sl ^ swz64(p)) << 3);
                    xh[i] = *reinterpret_cast<const f16x8*>(halo_hi + off);
                    xl[i] = *reinterpret_cast<const f16x8*>(halo_lo + off);
                }
                const int wbase = (wb * TAPS + t) * BN * 32 + r * 32 + ((g ^ swz64(r)) << 3);
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    const f16x8 ch = *reinterpret_cast<const f16x8*>(b_hi + wbase + j * 16 * 32);
                    const f16x8 cl = *reinterpret_cast<const f16x8*>(b_lo + wbase + j * 16 * 32);
                    cor[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cl, xh[0], cor[0][j], 0, 0, 0);
                    cor[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cl, xh[1], cor[1][j], 0, 0, 0);
                    acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch, xh[0], acc[0][j], 0, 0, 0);
                    acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch, xh[1], acc[1][j], 0, 0, 0);
                    cor[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch, xl[0], cor[0][j], 0, 0, 0);
                    cor[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch, xl[1], cor[1][j], 0, 0, 0);
                }
            }
        }
        STAMP(2)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // next stage's weights (issued a whole stage ago) and next chunk's halo loads
        STAMP(3)
        __syncthreads();
        STAMP(4)
        if (next_halo && !ABL(8)) {        // every wave has finished reading the old halo
#pragma unroll
            for (int k = 0; k < HALO_TPT; ++k) halo_store(k);
            __syncthreads();
            STAMP(5)
        }
    }

    // ---- epilogue ----
    float* orow[2];
    long long prow[2];                 // pixel index = row of the optional plane sink
    bool live[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int oy = oy0 + 2 * wave + i, ox = ox0 + r;
        live[i] = oy < a.H && ox < a.W;
        prow[i] = ((long long)img * a.H + (live[i] ? oy : 0)) * a.W + (live[i] ? ox : 0);
        orow[i] = a.out + prow[i] * a.out_ld;
    }
    // slopes of the plane sink's PReLU for all of this lane's column groups, fetched before the store loop (a load inside it
    // would put a vmcnt(0) round trip between consecutive stores)
    f32x4 psl[WN];
    if (a.out_hi && a.plane_prelu) {
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int co = n0 + 16 * j + 4 * g;
            psl[j] = *reinterpret_cast<const f32x4*>(a.plane_prelu + (co < a.Cout ? co : 0));       // padded to 32 by the host
        }
    } else {
#pragma unroll
        for (int j = 0; j < WN; ++j) psl[j] = (f32x4){1.f, 1.f, 1.f, 1.f};
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int cl = 16 * j + 4 * g;
        const int co = n0 + cl;
        const int nvalid = a.Cout - co;
        // bias / PReLU slopes from LDS (0 / 1 where absent): no global load and no wait in this loop, the stores stream out
        const f32x4 bv = *reinterpret_cast<const f32x4*>(cst + cl);
        const f32x4 pv = *reinterpret_cast<const f32x4*>(cst + BN + cl);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x4 v = acc[i][j] + cor[i][j] * LO_UNSCALE + bv;
            v.x = v.x > 0.f ? v.x : pv.x * v.x;
            v.y = v.y > 0.f ? v.y : pv.y * v.y;
            v.z = v.z > 0.f ? v.z : pv.z * v.z;
            v.w = v.w > 0.f ? v.w : pv.w * v.w;
            if (live[i] && !ABL(16)) {
                if (nvalid >= 4) {
                    *reinterpret_cast<f32x4*>(orow[i] + co) = v;
                } else if (nvalid > 0) {
                    orow[i][co] = v.x;
                    if (nvalid > 1) orow[i][co + 1] = v.y;
                    if (nvalid > 2) orow[i][co + 2] = v.z;
                }
                if (a.out_hi && nvalid > 0) {
                    // plane sink (uniform branch, two layers of the network): the value the NEXT layer reads, i.e. through that
                    // layer's leading PReLU; channels past Cout in this group of 4 are the planes' pad channels (zero weights,
                    // zero bias: v = 0) and are written as such
                    f32x4 u = v;
                    const f32x4 sl = psl[j];
                    u.x = u.x > 0.f ? u.x : sl.x * u.x;
                    u.y = u.y > 0.f ? u.y : sl.y * u.y;
                    u.z = u.z > 0.f ? u.z : sl.z * u.z;
                    u.w = u.w > 0.f ? u.w : sl.w * u.w;
                    if (nvalid < 4) {
                        u.y = nvalid > 1 ? u.y : 0.f;
                        u.z = nvalid > 2 ? u.z : 0.f;
                        u.w = 0.f;
                    }
                    const RowSink sink{nullptr, 0, a.out_hi, a.out_lo, a.plane_rows};
                    sink_store4(sink, prow[i], co, u);
                }
            }
        }
    }
    STAMP(6)
#ifdef ATMVFI_STAMP
    if (a.stamp && lane == 0) {
        unsigned long long* o = a.stamp + ((long long)blockIdx.x * 8 + wave) * 8;
        for (int k = 0; k < 8; ++k) o[k] = st_[k];
    }
#endif
}

template <int WN, int NWV, int TAPS>
int launch_row(const Conv3Dev& d, int ntiles, hipStream_t s) {
    constexpr int BN = 16 * WN;
    constexpr int TH = 2 * NWV, NPIX = HW_ * (TH + 2);
    const size_t lds = (size_t)(2 * NPIX * 32 + 2 * 2 * TAPS * BN * 32) * sizeof(_Float16) + epilogue_const_floats(BN) * sizeof(float);
    auto kern = conv3x3_f16x3_row_kernel<WN, NWV, TAPS>;
    const hipError_t attr_err = atmvfi::allow_dynamic_lds<conv3x3_f16x3_row_kernel<WN, NWV, TAPS>>(lds);
    ATMVFI_REQUIRE(attr_err == hipSuccess, ATMVFI_ELAUNCH, "conv3x3_f16x3_row: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    Conv3Dev ds = d;
    ds.nblocks = (ntiles + WN - 1) / WN;
    ds.tiles_y = (d.H + TH - 1) / TH;
    const long long sgroups = ((long long)d.N * d.tiles_x * ds.tiles_y + 7) / 8;
    ds.tchunk = (int)sgroups;
#if defined(ATMVFI_ABLATE) || defined(ATMVFI_STAMP)
    static const int legacy = [] { const char* e = getenv("ATMVFI_LEGACY_ORDER"); return e ? atoi(e) : 0; }();     // diagnostic builds only
    ds.legacy_order = legacy;
#else
    ds.legacy_order = 0;
#endif
    ATMVFI_REQUIRE(sgroups * 8 * ds.nblocks < (1LL << 31), ATMVFI_EINVAL, "conv3x3_f16x3_row: grid too large");
    dim3 grid((unsigned)(sgroups * 8 * ds.nblocks));
#ifdef ATMVFI_STAMP
    ds.stamp = g_stamp_buf;
#endif
    hipLaunchKernelGGL(kern, grid, dim3(64 * NWV), lds, s, ds);
    return atmvfi::check_launch("conv3x3_f16x3_row");
}

}  // namespace

int atmvfi::launch_conv3x3_row(const Conv3Dev& d, int ntiles, hipStream_t s) {
    // Tile width WN (n-tiles per workgroup) and schedule, chosen together: time ~ rounds x tile time.
    //   rounds    = ceil(workgroups / resident slots): one slot per CU for the 512-thread schedule, two for the half-size one.
    //               Counting rounds is what keeps small maps busy and whole: a 68x120 map with 768 channels is 240 workgroups at
    //               WN = 8 (one round on 256 CUs) but 320 at WN = 6 -- two rounds, the second a quarter full (0.65 against 0.42 ms).
    //   tile time ~ (WN + c0) [MFMA work ~ WN; halo staging, prologue and epilogue per tile ~ c0 = 2 n-tiles' worth, fitted to
    //               same-box sweeps of all (schedule, WN) pairs on the 1080p layers, tools/tune_conv3.py; with c0 = 1 the search
    //               picked 3-tile half workgroups for 389- and 576-wide layers and lost 10 %], x rel[WN] for the half schedule.
    //   rel[WN]   = measured time of a half-tile pair relative to a full tile at that width (1080p layers, same-box A/B, rounds
    //               factored out: 48- and 64-wide layers gain 10-13 %, 24-wide and 80-wide (5 n-tiles) ones lose 18-50 %,
    //               112-wide ones are even).  The half schedule overlaps two workgroups' DMA issue, barrier waits and halo
    //               conversion and quantises better on small images (8-row tiles), but streams the weights twice per 256 pixels
    //               and pays a barrier per tap; three taps per stage on the half tiles measured no better.
    // The caller may force either (atmvfi_conv3x3_f16x3's `schedule` / `wn` arguments: parity tests, sweeps); the library keeps no
    // override of its own.
    const int forced = d.force_schedule, wn_override = d.force_wn;
    static const float rel[9] = {1.f, 1.10f, 1.20f, 0.88f, 0.95f, 1.15f, 1.07f, 1.06f, 1.02f};     // re-measured with the pipelined fragment reads (tools/tune_conv3.py)
#if defined(ATMVFI_ABLATE) || defined(ATMVFI_STAMP)
    static const float c0 = [] { const char* e = getenv("ATMVFI_CONV3_C0"); return e ? (float)atof(e) : 2.0f; }();     // diagnostic builds only
#else
    constexpr float c0 = 2.0f;
#endif
    const int ncu = atmvfi::cu_count();
    const long long spatial_row = (long long)d.N * d.tiles_x * ((d.H + 15) / 16);
    const long long spatial_half = (long long)d.N * d.tiles_x * ((d.H + 7) / 8);
    int best = 1;
    bool half = false;
    float best_cost = 1e30f;
    for (int wn = 1; wn <= 8; ++wn) {
        if (wn_override > 0 && wn != wn_override) continue;
        const int nb = (ntiles + wn - 1) / wn;
        const float tile = (float)wn + c0;
        const float c_row = (float)((spatial_row * nb + ncu - 1) / ncu) * tile;
        const float c_half = (float)((spatial_half * nb + 2 * ncu - 1) / (2 * ncu)) * tile * rel[wn];
        if (forced != 1 && c_row <= best_cost) { best_cost = c_row; best = wn; half = false; }
        if (forced != 0 && c_half <= best_cost) { best_cost = c_half; best = wn; half = true; }
    }
#if defined(ATMVFI_ABLATE) || defined(ATMVFI_STAMP)
    static const bool verbose = getenv("ATMVFI_CONV3_VERBOSE") != nullptr;
    if (verbose) fprintf(stderr, "conv3x3 N%d H%d W%d Cin%d Cout%d -> WN %d %s (cost %.2f)\n", d.N, d.H, d.W, d.Cin, d.Cout, best, half ? "half" : "row", best_cost);
#endif
#define ATMVFI_C3_CASE(W) case W: return half ? launch_row<W, 4, 1>(d, ntiles, s) : launch_row<W, 8, 3>(d, ntiles, s);
    switch (best) {
        ATMVFI_C3_CASE(1) ATMVFI_C3_CASE(2) ATMVFI_C3_CASE(3) ATMVFI_C3_CASE(4) ATMVFI_C3_CASE(5) ATMVFI_C3_CASE(6) ATMVFI_C3_CASE(7)
        default: return half ? launch_row<8, 4, 1>(d, ntiles, s) : launch_row<8, 8, 3>(d, ntiles, s);
    }
#undef ATMVFI_C3_CASE
}

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
//   * the weights of one k-step (one tap x 32 channels) are one slot of a 4-deep LDS ring, filled three k-steps ahead;
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
// vmcnt bookkeeping is dynamic and per wave (waves issue different numbers of pieces): `issued` counts the wave's DMA
// instructions, mark[k-step & 3] remembers the count after the pieces of that k-step (and every halo piece before them) went out;
// the wait is vmcnt(issued - mark), through a switch on a wave-uniform value (s_waitcnt only takes an immediate).
#include "conv3_common.h"

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
    int tiles_x, tiles_y, nblocks, tchunk;
};

constexpr int NB_W = 4;                       // weight ring slots (k-steps)
constexpr int LOOKAHEAD = 3;                  // k-steps between a slot's DMA issue and its first read
constexpr int HALO_PIX = HW_ * HW_;           // 324
constexpr int HALO_PLANE_PIECES = (HALO_PIX + 15) / 16;     // 21 one-KiB pieces (16 pixel rows x 64 B) per plane, 12 pad rows
constexpr int HALO_LO = HALO_PLANE_PIECES * 1024;            // byte offset of the lo plane inside a halo buffer
constexpr int HALO_BYTES = 2 * HALO_LO;

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// vmcnt(min(n, 7)) for a wave-uniform n = operations that may stay outstanding (s_waitcnt only takes an immediate; waiting for a
// smaller count than allowed is always safe).  A three-level decision tree: the switch hipcc builds costs ~40 scalar instructions.
__device__ __forceinline__ void wait_vm_dyn(int n) {
    if (n >= 4) {
        if (n >= 6) { if (n >= 7) wait_vm<7>(); else wait_vm<6>(); }
        else { if (n == 5) wait_vm<5>(); else wait_vm<4>(); }
    } else {
        if (n >= 2) { if (n == 3) wait_vm<3>(); else wait_vm<2>(); }
        else { if (n == 1) wait_vm<1>(); else wait_vm<0>(); }
    }
}
__device__ __forceinline__ void dma16(const unsigned char* src, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int WN>
__global__ __launch_bounds__(512, 1) void conv3x3_planes_kernel(const Conv3PDev a) {
    constexpr int BN = 16 * WN;
    constexpr int WSLOT = 2 * BN * 64;                  // bytes of one ring slot: [hi BN rows][lo BN rows] x 64 B
    constexpr int SW = (2 * WN + 7) / 8;                // weight pieces per wave and k-step (some waves one fewer)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned halo0 = lds_offset(smem);                        // two halo buffers
    const unsigned ring0 = halo0 + 2 * HALO_BYTES;                  // NB_W weight slots
    float* cst = reinterpret_cast<float*>(smem + 2 * HALO_BYTES + NB_W * WSLOT);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int r = lane & 15;
    const int g = lane >> 4;

    // XCD-aware tile order (conv3x3_f16x3_row.hip): column blocks of one tile back to back on one XCD, each XCD walking a
    // contiguous eighth of the tiles in groups of 8 tile rows, column by column
    const int slot = blockIdx.x >> 3;
    const int sgrp = slot / a.nblocks;
    const int nblk = slot - sgrp * a.nblocks;
    int L = (blockIdx.x & 7) * a.tchunk + sgrp;
    const int per_img = a.tiles_x * a.tiles_y;
    if (L >= a.N * per_img) return;
    const int img = L / per_img;
    L -= img * per_img;
    const int tgrp = L / (8 * a.tiles_x);
    const int rem = L - tgrp * 8 * a.tiles_x;
    const int rows_here = (a.tiles_y - 8 * tgrp) < 8 ? a.tiles_y - 8 * tgrp : 8;
    const int txb = rem / rows_here;
    const int tyb = 8 * tgrp + (rem - txb * rows_here);
    const int ox0 = txb * TW, oy0 = tyb * 16;
    const int n0 = nblk * BN;

    // ---- halo pieces of this wave: k = wave + 8 s, s = 0..5 (k < 42): pieces 0..20 = hi plane, 21..41 = lo plane, each plane
    // a linear image of 336 pixel rows x 64 B (324 used).  Lane -> pixel row hp = 16 (k % 21) + lane / 4, physical 16-byte slot
    // lane & 3; the slot swizzle goes on the SOURCE (the DMA destination is lane-linear).  hoff = byte offset from the plane base
    // of the chunk; pixels outside the image (and the 12 pad rows) read the planes' zero row N*H*W.
    unsigned hoff[6];
    const long long zero_row = (long long)a.N * a.H * a.W;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int k = wave + 8 * s;
        const int kp = k >= HALO_PLANE_PIECES ? k - HALO_PLANE_PIECES : k;
        const int hp = 16 * kp + (lane >> 2);
        const int hy = hp / HW_, hx = hp - hy * HW_;
        const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
        const bool ok = hp < HALO_PIX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const int ls = (lane & 3) ^ swz64(hp);
        const long long row = ok ? ((long long)img * a.H + iy) * a.W + ix : zero_row;
        hoff[s] = (unsigned)(row * 64 + ls * 16);
    }
    const long long chunk_halves = a.in_rows * 32;
    // ---- weight pieces of this wave: idx = wave + 8 s < 2 WN -> plane idx / WN, row group idx % WN (16 rows x 64 B = one
    // contiguous KiB of the k-step-major planes); the lane part of the address is the same for every piece
    const unsigned wlane = (unsigned)((lane >> 2) * 64 + (((lane & 3) ^ swz64(lane >> 2)) << 4));
    const long long step_halves = (long long)a.wrows * 32;

    const int nfull = a.cf >> 5;
    const int nchunks = nfull + (a.tail ? 1 : 0);
    const int nk = 9 * nfull + (a.tail ? 3 : 0);

    int issued = 0;                       // DMA instructions this wave has issued
    int mark[4] = {0, 0, 0, 0};           // issued count right after the pieces of k-step (u & 3)
    auto issue_weights = [&](int u) {     // weights of k-step u -> ring slot u % NB_W
        unsigned char* dst = smem + 2 * HALO_BYTES + (u & (NB_W - 1)) * WSLOT;
#pragma unroll
        for (int s = 0; s < SW; ++s) {
            const int idx = wave + 8 * s;
            if (idx < 2 * WN) {
                const int plane = idx >= WN ? 1 : 0;
                const int j = idx - plane * WN;
                int rg = n0 + 16 * j;
                if (rg >= a.wrows) rg = a.wrows - 16;                    // row groups past the packed rows: columns never stored
                const _Float16* base = (plane ? a.w_lo : a.w_hi) + u * step_halves + (long long)rg * 32;      // wave-uniform
                dma16(reinterpret_cast<const unsigned char*>(base) + wlane, dst + (plane * BN + 16 * j) * 64);
                ++issued;
            }
        }
        mark[u & 3] = issued;
    };
    auto issue_halo = [&](int chunk, auto sc) {       // halo piece wave + 8 S of `chunk` -> buffer chunk & 1
        constexpr int S = decltype(sc)::value;
        const int k = wave + 8 * S;
        if (k < 2 * HALO_PLANE_PIECES) {
            const _Float16* base = (k >= HALO_PLANE_PIECES ? a.in_lo : a.in_hi) + chunk * chunk_halves;          // wave-uniform
            dma16(reinterpret_cast<const unsigned char*>(base) + hoff[S], smem + (chunk & 1) * HALO_BYTES + k * 1024);
            ++issued;
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

    // tail k-steps: lane group g reads slot 0 of the halo pixel of tap 4t + g (taps 9..11 meet zero weights: tap 8 again)
    int dtail = 0;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int tap = (4 * t + g) < 9 ? 4 * t + g : 8;
        const int ty = tap / 3;
        dtail |= (ty * HW_ + (tap - 3 * ty)) << (8 * t);
    }

    // ---- prologue: epilogue constants, halo of chunk 0, weights of k-steps 0 .. LOOKAHEAD-1 ----
    dma_epilogue_consts<BN>(a.bias, a.prelu, n0, cst, wave, lane, [&](int col) { return col < a.Cout ? col : -1; });
    ++issued;
    static_for<0, 6>([&](auto sc) { issue_halo(0, sc); });
#pragma unroll
    for (int u = 0; u < LOOKAHEAD; ++u)
        if (u < nk) issue_weights(u);
    wait_vm_dyn(issued - mark[0]);
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();          // the second group runs one phase behind

    f16x8 xh[2], xl[2], wh[WN], wl[WN];
    const unsigned wfrag = ring0 + (unsigned)(r * 64 + ((g ^ swz64(r)) << 4));       // swz64(16 j + r) == swz64(r)
    const int prow = 2 * wave * HW_ + r;                                                // halo pixel of (row 2 wave, column r), tap (0,0)

    // One k-step of one wave.  T = tap (regular chunk) or tail step; u = global k-step index, chunk = its 32-channel chunk.
    auto kstep = [&](auto tc, auto tailc, int u, int chunk) {
        constexpr int T = decltype(tc)::value;
        constexpr bool TAIL = decltype(tailc)::value;
        // ---------------- read phase ----------------
        const unsigned hb = halo0 + (unsigned)(chunk & 1) * HALO_BYTES;
        int pr = prow;
        asm volatile("" : "+v"(pr));       // opaque: keeps hipcc from hoisting the 18 fragment addresses of a chunk out of the loop
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = pr + i * HW_ + (TAIL ? ((dtail >> (TAIL ? 8 * T : 0)) & 0xff) : (T / 3) * HW_ + T % 3);
            const int sl = TAIL ? 0 : g;
            const unsigned addr = hb + (unsigned)(p * 64 + ((sl ^ swz64(p)) << 4));
            lds_read16<0>(xh[i], addr);
            lds_read16<HALO_LO>(xl[i], addr);
        }
        const unsigned wa = wfrag + (unsigned)(u & (NB_W - 1)) * WSLOT;
        static_for<0, WN>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            lds_read16<j * 1024>(wh[j], wa);
            lds_read16<j * 1024 + BN * 64>(wl[j], wa);
        });
        // DMA: halo of the next chunk (k-steps 0..5 of a regular chunk), weights LOOKAHEAD k-steps ahead
        if constexpr (!TAIL && T < 6) {
            if (chunk + 1 < nchunks) issue_halo(chunk + 1, std::integral_constant<int, T>{});
        }
        if (u + LOOKAHEAD < nk) issue_weights(u + LOOKAHEAD);
        // own pieces of k-step u+1 landed (younger ones stay in flight), own fragment reads complete
        if (u + 1 < nk) wait_vm_dyn(issued - mark[(u + 1) & 3]);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(xh[0]), "+v"(xh[1]), "+v"(xl[0]), "+v"(xl[1]));
#pragma unroll
        for (int j = 0; j < WN; ++j) asm volatile("" : "+v"(wh[j]), "+v"(wl[j]));
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- MFMA phase ----------------
        __builtin_amdgcn_s_setprio(1);
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
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    for (int c = 0; c < nfull; ++c)
        static_for<0, 9>([&](auto tc) { kstep(tc, std::false_type{}, 9 * c + decltype(tc)::value, c); });
    if (a.tail)
        static_for<0, 3>([&](auto tc) { kstep(tc, std::true_type{}, 9 * nfull + decltype(tc)::value, nfull); });
    if (grp == 0) __builtin_amdgcn_s_barrier();          // same number of barriers for both groups

    // ---- epilogue (as conv3x3_f16x3_row.hip) ----
    float* orow[2];
    long long prow_o[2];
    bool live[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int oy = oy0 + 2 * wave + i, ox = ox0 + r;
        live[i] = oy < a.H && ox < a.W;
        prow_o[i] = ((long long)img * a.H + (live[i] ? oy : 0)) * a.W + (live[i] ? ox : 0);
        orow[i] = a.out ? a.out + prow_o[i] * a.out_ld : nullptr;
    }
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
        const f32x4 bv = *reinterpret_cast<const f32x4*>(cst + cl);
        const f32x4 pv = *reinterpret_cast<const f32x4*>(cst + BN + cl);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x4 v = acc[i][j] + cor[i][j] * LO_UNSCALE + bv;
            v.x = v.x > 0.f ? v.x : pv.x * v.x;
            v.y = v.y > 0.f ? v.y : pv.y * v.y;
            v.z = v.z > 0.f ? v.z : pv.z * v.z;
            v.w = v.w > 0.f ? v.w : pv.w * v.w;
            if (live[i]) {
                if (a.out) {
                    if (nvalid >= 4) {
                        *reinterpret_cast<f32x4*>(orow[i] + co) = v;
                    } else if (nvalid > 0) {
                        orow[i][co] = v.x;
                        if (nvalid > 1) orow[i][co + 1] = v.y;
                        if (nvalid > 2) orow[i][co + 2] = v.z;
                    }
                }
                if (a.out_hi && nvalid > 0) {
                    f32x4 u = v;
                    const f32x4 sl = psl[j];
                    u.x = u.x > 0.f ? u.x : sl.x * u.x;
                    u.y = u.y > 0.f ? u.y : sl.y * u.y;
                    u.z = u.z > 0.f ? u.z : sl.z * u.z;
                    u.w = u.w > 0.f ? u.w : sl.w * u.w;
                    if (nvalid < 4) {          // channels past Cout in this group of 4: the planes' pad channels, written as zero
                        u.y = nvalid > 1 ? u.y : 0.f;
                        u.z = nvalid > 2 ? u.z : 0.f;
                        u.w = 0.f;
                    }
                    const RowSink sink{nullptr, 0, a.out_hi, a.out_lo, a.plane_rows};
                    sink_store4(sink, prow_o[i], a.out_c0 + co, u);
                }
            }
        }
    }
}

template <int WN>
int launch_planes(const Conv3PDev& d, int ntiles, hipStream_t s) {
    constexpr int BN = 16 * WN;
    const size_t lds = (size_t)2 * HALO_BYTES + (size_t)NB_W * 2 * BN * 64 + epilogue_const_floats(BN) * sizeof(float);
    auto kern = conv3x3_planes_kernel<WN>;
    static const hipError_t attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    ATMVFI_REQUIRE(attr_err == hipSuccess, ATMVFI_ELAUNCH, "conv3x3_planes: hipFuncSetAttribute: %s", hipGetErrorString(attr_err));
    Conv3PDev ds = d;
    ds.nblocks = (ntiles + WN - 1) / WN;
    ds.tiles_y = (d.H + 15) / 16;
    const long long sgroups = ((long long)d.N * d.tiles_x * ds.tiles_y + 7) / 8;
    ds.tchunk = (int)sgroups;
    ATMVFI_REQUIRE(sgroups * 8 * ds.nblocks < (1LL << 31), ATMVFI_EINVAL, "conv3x3_planes: grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)(sgroups * 8 * ds.nblocks)), dim3(512), lds, s, ds);
    return atmvfi::check_launch("conv3x3_planes");
}

}  // namespace

extern "C" int atmvfi_conv3x3_planes(const void* in_hi, const void* in_lo, int64_t in_rows, int N, int H, int W, int Cin, const void* w_hi,
                                      const void* w_lo, int Cout, float* out, int out_ld, const float* bias, const float* prelu,
                                      void* out_hi, void* out_lo, int64_t plane_rows, int out_c0, const float* plane_prelu, int wn,
                                      void* stream) {
    ATMVFI_REQUIRE(in_hi && in_lo && w_hi && w_lo && (out || out_hi), ATMVFI_EINVAL, "conv3x3_planes: null pointer");
    ATMVFI_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, ATMVFI_EINVAL, "conv3x3_planes: bad shape");
    ATMVFI_REQUIRE(in_rows > (int64_t)N * H * W && in_rows * 64 < (1ll << 32), ATMVFI_EINVAL,
                   "conv3x3_planes: in_rows %lld must exceed N*H*W (the zero row) and in_rows * 64 must fit 32 bits", (long long)in_rows);
    ATMVFI_REQUIRE(atmvfi::aligned16(in_hi) && atmvfi::aligned16(in_lo) && atmvfi::aligned16(w_hi) && atmvfi::aligned16(w_lo) &&
                       (!bias || atmvfi::aligned16(bias)) && (!prelu || atmvfi::aligned16(prelu)),
                   ATMVFI_EALIGN, "conv3x3_planes: pointers (incl. bias/prelu) must be 16-byte aligned");
    if (out)
        ATMVFI_REQUIRE(atmvfi::aligned16(out) && out_ld % 4 == 0 && out_ld >= atmvfi::round_up(Cout, 4), ATMVFI_EALIGN,
                       "conv3x3_planes: fp32 output must be 16-byte aligned with ld %% 4 == 0 covering the channels");
    ATMVFI_REQUIRE((out_hi == nullptr) == (out_lo == nullptr), ATMVFI_EINVAL, "conv3x3_planes: plane sink needs both planes");
    if (out_hi) {
        ATMVFI_REQUIRE(plane_rows >= (int64_t)N * H * W && out_c0 >= 0 && out_c0 % 4 == 0, ATMVFI_EINVAL,
                       "conv3x3_planes: plane sink needs plane_rows >= N*H*W and a channel offset that is a multiple of 4");
        ATMVFI_REQUIRE(atmvfi::aligned16(out_hi) && atmvfi::aligned16(out_lo) && (!plane_prelu || atmvfi::aligned16(plane_prelu)),
                       ATMVFI_EALIGN, "conv3x3_planes: plane sink pointers must be 16-byte aligned");
    }
    ATMVFI_REQUIRE(wn >= 0 && wn <= 8, ATMVFI_EINVAL, "conv3x3_planes: wn 0 (auto) or 1..8");
    Conv3PDev d;
    d.in_hi = (const _Float16*)in_hi; d.in_lo = (const _Float16*)in_lo; d.in_rows = in_rows;
    d.N = N; d.H = H; d.W = W; d.Cin = Cin;
    d.w_hi = (const _Float16*)w_hi; d.w_lo = (const _Float16*)w_lo;
    d.wrows = atmvfi::round_up(Cout, 16);
    const int t = Cin % 32;
    if (t >= 1 && t <= 8) { d.cf = Cin - t; d.tail = t; } else { d.cf = atmvfi::round_up(Cin, 32); d.tail = 0; }
    d.Cout = Cout; d.out = out; d.out_ld = out_ld; d.bias = bias; d.prelu = prelu;
    d.out_hi = (_Float16*)out_hi; d.out_lo = (_Float16*)out_lo; d.plane_rows = plane_rows; d.out_c0 = out_c0; d.plane_prelu = plane_prelu;
    d.tiles_x = (W + TW - 1) / TW;
    d.tiles_y = 0; d.nblocks = 0; d.tchunk = 0;
    const int ntiles = (Cout + 15) / 16;
    // tile width: rounds x (WN + c0), one workgroup per CU (conv3x3_f16x3_row.hip's cost model, row schedule)
    int best = wn;
    if (best == 0) {
        const int ncu = atmvfi::cu_count();
        const long long spatial = (long long)N * d.tiles_x * ((H + 15) / 16);
        float best_cost = 1e30f;
        for (int w = 1; w <= 8; ++w) {
            const int nb = (ntiles + w - 1) / w;
            const float c = (float)((spatial * nb + ncu - 1) / ncu) * ((float)w + 2.0f);
            if (c <= best_cost) { best_cost = c; best = w; }
        }
    }
    hipStream_t s = (hipStream_t)stream;
    switch (best) {
        case 1: return launch_planes<1>(d, ntiles, s);
        case 2: return launch_planes<2>(d, ntiles, s);
        case 3: return launch_planes<3>(d, ntiles, s);
        case 4: return launch_planes<4>(d, ntiles, s);
        case 5: return launch_planes<5>(d, ntiles, s);
        case 6: return launch_planes<6>(d, ntiles, s);
        case 7: return launch_planes<7>(d, ntiles, s);
        default: return launch_planes<8>(d, ntiles, s);
    }
}

// DEFERRED-EPILOGUE 3x3 split-plane convolution kernel (gfx950): conv3x3_planes_de.hip.
//
// conv3x3_planes_kernel (conv3x3_planes.hip) stops the matrix pipes at every tile boundary: both wave groups run their epilogues --
// VALU-bound, 4.7 k + 7.3 k cycles of a 55 k-cycle 112-column tile, a third of a 64-column tile of 18 k-steps (tools/stamp_conv3p.py) --
// and nothing else can run, because all 256 registers of a wave hold the two accumulator sets of the f16x3 arithmetic (acc: hi x hi,
// cor: the two cross products, folded as acc + cor / 1024).  Here:
//   * the three products of a k-step go into ONE fp32 accumulator: the activation fragments of the cross products are scaled by 2^-10
//     in registers first (v_pk_mul_f16 in the read phase; fp16 subnormals for |x| < 2^-4, which gfx950's matrix cores keep: tools/probes/
//     mfma_denorm_probe.hip; tools/sim_single_acc.py).  Not bit-identical to the two-accumulator kernel; same tolerance class;
//   * the freed registers hold the PREVIOUS tile's sums (pend): the first 2 WN MFMAs of a tile start the new sums from C = 0 in fresh
//     registers (hipcc renames: no copy is executed), and that tile's epilogue runs as ATOMS of at most two plain vector instructions,
//     each behind one MFMA, dealt evenly over the MFMA slots of the new tile's first chunk (9 k-steps);
//   * the k-loop runs from tile to tile with no boundary: no extra barrier, no drain;
//   * the kernel arguments are re-read from the kernarg segment by the per-tile code instead of living in scalar registers across the
//     k-loop (as a by-value struct they made hipcc spill 120-150 scalars to vector-register lanes, and every use was a v_readlane in the
//     middle of an MFMA phase); the k-loop's scalar bookkeeping is pinned to the read phase.
// What it buys, and why not more (profiles/r05_de_experiments.txt, r05_de_v5_phase_stamps.txt): 64 -> 64 at 1088 x 1920 0.463 -> 0.436 ms,
// 101 -> 101 1.073 -> 1.020; the bare k-loop (no epilogue at all) runs 0.369.  A v_mfma_f32_16x16x32 holds its SIMD's vector issue for 8
// of its 16 cycles (MI355X_MICROARCH.md), and that budget is shared by BOTH waves of the SIMD: a vector instruction costs ~3 cycles of
// matrix time even as one of two per MFMA gap (an MFMA phase with atoms: 1 070-1 230 ticks against 885-900), the same units as one block
// in the wave's read phase cost more (0.446), and the stores cost what the memory system takes to accept them (0.369 -> 0.419 with
// EXEC = 0 stores, 0.430 storing every tile to one KiB, 0.456 for real), wherever they are issued.
// Counted vmcnt waits: the immediates count the LDS-DMA pieces issued behind the awaited one AND the plane stores of the epilogue atoms
// (vmcnt counts loads, stores and LDS-DMA together, in issue order: MI355X_MICROARCH.md; hipcc's own counted waits rely on it).
//
// ONE output form (the launcher, conv3x3_planes.hip, keeps everything else on conv3x3_planes_kernel): bias, PReLU (absent: slope 1), one
// raw plane sink -- the encoder, the motion MLPs, the refiner, the decoder's first convs.  The decoder's second convs (fp32 rows for five
// channels, a sink through its own PReLU, a second raw sink: 60-110 instructions per unit) were built and measured neutral to +1.5 %
// slower than the two-accumulator kernel (1.163 -> 1.180 ms, 1.036 -> 1.041): not kept.
#include "conv3p.h"

// Experiment builds only (tools/lib/, never the product: wrong results): ATMVFI_DE_EXP bit 0 = no epilogue units at all (the bare
// k-loop), bit 1 = units without their stores, bit 2 = no raised priority in the MFMA phases.
#ifndef ATMVFI_DE_EXP
#define ATMVFI_DE_EXP 0
#endif

// Diagnostic build only (make stamp; tools/stamp_conv3p_de.py): four s_memtime per k-step of a workgroup's SECOND tile -- start / end of the
// read phase, start / end of the MFMA phase -- kept in the lanes of two vector registers and written out at the end.
#ifdef ATMVFI_STAMP
#define DSTAMP(idx)                                                                                                                 \
    if (seq == 1) {                                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                                          \
        const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime();                                                                 \
        if ((idx) < 64) asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(stv0) : "s"(t_), "n"((idx) & 63));                           \
        else asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(stv1) : "s"(t_), "n"((idx) & 63));                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                                          \
    }
#else
#define DSTAMP(idx)
#endif

namespace {
using namespace atmvfi;

// ---- The epilogue of a tile = NUNIT units (n-tile pair jp, pixel row i), each a list of ATOMS of at most two plain vector instructions
// (MI355X_MICROARCH.md, constants: a v_mfma_f32_16x16x32 holds its SIMD's vector issue for 8 of its 16 cycles, so two 4-cycle
// instructions per MFMA gap are nearly free and everything beyond costs its full issue time; packed fp32 -- v_pk_add / v_pk_mul -- is an
// anti-lever beside MFMAs: the arithmetic below is written per component, and these files are compiled without the SLP vectorizer).
// Atom codes (k = position in the unit):
//   10 + 10 t + k   n-tile t of the pair: k = 0 its constants (bias, slope), 1, 2: + bias (two values each), 3, 4: slope x value,
//                   5..8: PReLU of value 0..3
//   60 + k          the plane sink: k = 3 p + {0, 1, 2}: the split of value pair p = 0..3 in three parts, 12, 13: cross-half swaps of the
//                   hi / lo dwords, 14: store mask and offset, 15: the two stores
constexpr int DE_NATOMS = 34;
constexpr int de_atom(int k) { return k < 9 ? 10 + k : k < 18 ? 20 + (k - 9) : 60 + (k - 18); }
constexpr bool de_is_store(int atom) { return atom == 75; }
// The atoms of a tile's units are dealt evenly over the MFMA slots of the next tile's first chunk (k-step 0 from its third pass on: its
// first 2 WN MFMAs carry the copy acc -> pend).  The counted waits of the FOLLOWING chunk's first LA - 1 k-steps -- one code body for
// every later chunk, which cannot count stores -- therefore also wait for the stores of this chunk's last k-steps, once per tile.
constexpr int de_nslot(int wn) { return 9 * 6 * wn - 2 * wn; }
constexpr int de_qtot(int wn) { return 2 * ((wn + 1) / 2) * DE_NATOMS; }
// carrying-slot number of MFMA m of k-step t (negative: carries nothing), and the first atom of slot gs
constexpr int de_slot(int wn, int t, int m) { return t == 0 ? m - 2 * wn : 4 * wn + (t - 1) * 6 * wn + m; }
constexpr int de_q0(int wn, int gs) { return (gs * de_qtot(wn) + de_nslot(wn) - 1) / de_nslot(wn); }
// plane-sink store instructions (two per store atom: issued by every wave, masked or not) in the MFMA phase of k-step t
constexpr int de_kstep_stores(int wn, int t) {
    int n = 0;
    if (t < 0 || t >= 9) return 0;
    for (int m = 0; m < 6 * wn; ++m) {
        const int gs = de_slot(wn, t, m);
        if (gs < 0) continue;
        const int qa = de_q0(wn, gs), qe = de_q0(wn, gs + 1), qb = qe < de_qtot(wn) ? qe : de_qtot(wn);
        for (int q = qa; q < qb; ++q)
            if (de_is_store(de_atom(q % DE_NATOMS))) n += 2;
    }
    return n;
}
// ... behind the weights awaited in the read phase of k-step T of a tile's first chunk (issued in read phase T + 1 - LA): those of the
// MFMA phases T + 1 - LA .. T - 1
constexpr int de_stores_behind(int wn, int T, int la) {
    int n = 0;
    for (int t = (T + 1 - la > 0 ? T + 1 - la : 0); t < T; ++t) n += de_kstep_stores(wn, t);
    return n;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const Conv3PDev __attribute__((address_space(4)))* KArgs;

// ---- masked stores: EXEC = lanes whose mask word is not zero, the store(s), EXEC = all lanes again.  (s_nop 3: a scalar operand that hipcc
// reloads from a spill lane -- v_readlane -- right in front of the block is read by a memory instruction no earlier than 5 wait states on.)
__device__ __forceinline__ void mstore_pair16(unsigned mask, unsigned off, const u32x4& d0, const u32x4& d1, void* b0, void* b1) {
    if constexpr ((ATMVFI_DE_EXP & 2) != 0) return;
    asm volatile("v_cmpx_ne_u32_e32 vcc, 0, %0\n\t"
                 "s_nop 3\n\t"
                 "global_store_dwordx4 %1, %2, %4\n\t"
                 "global_store_dwordx4 %1, %3, %5\n\t"
                 "s_mov_b64 exec, -1"
                 :
                 : "v"(mask), "v"(off), "v"(d0), "v"(d1), "s"(b0), "s"(b1)
                 : "vcc", "memory");
}
// PReLU without VCC: d = v < 0 (sign bit) ? t : v with t = slope * v.  Bit for bit the select v > 0 ? v : slope * v except for the sign of
// a zero result (v = +0 with a negative slope); two instructions per value (v_ashrrev_i32, v_bfi_b32), no wait states.
__device__ __forceinline__ float sel_neg(float v, float t) {
    const unsigned vb = __builtin_bit_cast(unsigned, v), tb = __builtin_bit_cast(unsigned, t);
    const unsigned m = (unsigned)((int)vb >> 31);
    unsigned d;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(m), "v"(tb), "v"(vb));        // (m & t) | (~m & v): hipcc makes v_max_i32 + v_and_or_b32 of the C form
    return __builtin_bit_cast(float, d);
}
// common.h's split_pair in three parts of at most two plain instructions (same roundings, same bits: hi = fp16 pair of (x, y);
// lo' = fp16(fma(hi, -1024, x * 1024)) per value, x * 1024 exact):  A: hi, x * 1024  |  B: y * 1024, lo'(x)  |  C: lo'(y)
__device__ __forceinline__ void split_a(float x, float y, unsigned& hi, float& sx) {
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){x, y}, f16x2));
    sx = x * 1024.0f;
}
__device__ __forceinline__ void split_b(float y, unsigned hi, float sx, float& sy, unsigned& lo) {
    const float k = 1024.0f;
    sy = y * 1024.0f;
    asm("v_fma_mixlo_f16 %0, %1, -%2, %3 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(k), "v"(sx));
}
__device__ __forceinline__ void split_c(unsigned hi, float sy, unsigned& lo) {
    const float k = 1024.0f;
    asm("v_fma_mixhi_f16 %0, %1, -%2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(k), "v"(sy));
}
template <typename T>
__device__ __forceinline__ void pin_s(T& x) { asm volatile("" : "+s"(x)); }

template <int WN>
__global__ __launch_bounds__(512, 1) void conv3x3_planes_de_kernel(const Conv3PDev a_by_value) {
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

    // The kernel's arguments are read from the kernarg segment where they are needed (scalar loads through a laundered pointer: the
    // per-tile code re-reads them) instead of all ~75 dwords living in scalar registers across the k-loop: as a by-value struct they
    // made hipcc spill 120-150 scalars to vector-register lanes, and every use -- the store bases of the epilogue units among them --
    // was a v_readlane in the middle of an MFMA phase.
    (void)a_by_value;
    const KArgs ka0 = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    auto fresh = [&]() -> KArgs { KArgs p = ka0; asm volatile("" : "+s"(p)); return p; };

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int r = lane & 15;
    const int g = lane >> 4;

    // (no split-K, no fused read-out in this kernel: the launcher keeps those on conv3x3_planes_kernel)
    int nfull, ktail;
    {
        const KArgs a = fresh();
        nfull = a->cf >> 5;          // >= 1 (launcher)
        ktail = a->tail;
    }
    // PERSISTENT GRID over the XCD-aware tile order: see conv3x3_planes_kernel
    const int grid = gridDim.x;
    auto fdiv = [](int n, unsigned m, unsigned sh) -> int { return (int)((__umulhi((unsigned)n, m) + (unsigned)n) >> sh); };
    // (the per-tile code of one place reads the arguments through ONE laundered pointer: hipcc then batches the scalar loads and their
    // latency -- ~200 cycles from the kernarg segment -- is paid once, not once per helper)
    auto decode = [&](const KArgs a, int v, int& t_img, int& t_ox0, int& t_oy0, int& t_n0) -> bool {
        const int per_img = a->tiles_x * a->tiles_y;
        const int slot = v >> 3;
        const int sgrp = fdiv(slot, a->dm_nblocks, a->ds_nblocks);
        const int nblk = slot - sgrp * a->nblocks;
        int L = (v & 7) * a->tchunk + sgrp;
        if (v >= a->vblocks || L >= a->N * per_img) return false;
        t_img = fdiv(L, a->dm_perimg, a->ds_perimg);
        L -= t_img * per_img;
        const int tgrp = fdiv(L, a->dm_grp, a->ds_grp);              // groups of 8 tile rows
        const int rem = L - tgrp * 8 * a->tiles_x;
        const bool full = a->tiles_y - 8 * tgrp >= 8;
        const int rows_here = full ? 8 : a->tiles_y - 8 * tgrp;     // (the last group of a map may be shorter: tiles_y mod 8 rows)
        const int txb = full ? rem >> 3 : fdiv(rem, a->dm_rows, a->ds_rows);
        const int tyb = 8 * tgrp + (rem - txb * rows_here);
        t_ox0 = txb * TW;
        t_oy0 = tyb * 16;
        t_n0 = nblk * BN;
        return true;
    };
    int vb = blockIdx.x;
    int img, ox0, oy0, n0;                       // the tile of the MFMAs
    const KArgs a0 = fresh();
    if (!decode(a0, vb, img, ox0, oy0, n0)) return;
    int nimg = 0, nox0 = 0, noy0 = 0, nn0 = 0;   // this workgroup's next tile
    bool has_next = decode(a0, vb + grid, nimg, nox0, noy0, nn0);

    // ---- halo pieces of this wave (see conv3x3_planes_kernel): k = wave + 8 s, s = 0..5; hoff = byte offset from the plane base of the
    // chunk; pixels outside the image (and the 12 pad rows) read the planes' zero row N*H*W
    unsigned hoff[6];
    // The lane's part of hoff is the same for every tile whose halo lies inside the image: (hy W + hx) 64 + slot 16 from the plane offset of
    // the halo's first pixel (pad rows: slot 16 from the zero row).  Kept in registers (hrel, hpad: bit s = piece s of this lane is a pad
    // row); a tile away from the image border -- all but the perimeter -- then costs one instruction per piece instead of ~15, issued
    // behind the last MFMA of the chunk in front of the switch (kstep).
    unsigned hrel[6];
    const int H = a0->H, W = a0->W;                       // (kept in scalar registers: read by every tile's halo set-up)
    const long long zero_row = (long long)a0->N * H * W;
    {
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            int k = wave + 8 * s;
            if (k >= 2 * HALO_PLANE_PIECES) k -= 8;
            const int kp = k >= HALO_PLANE_PIECES ? k - HALO_PLANE_PIECES : k;
            const int hp = 16 * kp + (lane >> 2);
            const int hy = hp / HW_, hx = hp - hy * HW_;
            const int ls = (lane & 3) ^ swz64(hp);
            const bool pad = hp >= HALO_PIX;         // (the 12 pad rows of a plane's LDS image are never read: in the fast path they load the halo's first pixel)
            hrel[s] = (pad ? 0u : (unsigned)(hy * W + hx) * 64u) + (unsigned)ls * 16u;
        }
    }
    // plane offset of the first halo pixel of a tile whose halo lies inside the image (the fast path's scalar part)
    auto halo_inside = [&](int t_ox0, int t_oy0) -> bool { return t_oy0 >= 1 && t_oy0 + 16 < H && t_ox0 >= 1 && t_ox0 + 16 < W; };
    auto halo_base = [&](int t_img, int t_ox0, int t_oy0) -> unsigned { return (unsigned)((((long long)t_img * H + t_oy0 - 1) * W + t_ox0 - 1) * 64); };
    auto setup_halo = [&](int t_img, int t_ox0, int t_oy0) {
        if (halo_inside(t_ox0, t_oy0)) {
            const unsigned tb = halo_base(t_img, t_ox0, t_oy0);
#pragma unroll
            for (int s = 0; s < 6; ++s) hoff[s] = tb + hrel[s];
            return;
        }
        int ln = lane;
        asm volatile("" : "+v"(ln));             // (laundered: hoisted out of the tile loop these are 30 more live registers)
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            int k = wave + 8 * s;
            if (k >= 2 * HALO_PLANE_PIECES) k -= 8;      // waves 2..7 have no sixth piece: they send their fifth twice
            const int kp = k >= HALO_PLANE_PIECES ? k - HALO_PLANE_PIECES : k;
            const int hp = 16 * kp + (ln >> 2);
            const int hy = hp / HW_, hx = hp - hy * HW_;
            const int iy = t_oy0 - 1 + hy, ix = t_ox0 - 1 + hx;
            const bool ok = hp < HALO_PIX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const int ls = (ln & 3) ^ swz64(hp);
            const long long row = ok ? ((long long)t_img * H + iy) * W + ix : zero_row;
            hoff[s] = (unsigned)(row * 64 + ls * 16);
        }
    };
    setup_halo(img, ox0, oy0);
    // ---- DMA schedule: branch-free and the same for every wave, so that the vmcnt waits are plain immediates (conv3x3_planes_kernel)
    const int wrow = 8 * ((lane >> 4) & 1) + 4 * (lane >> 5) + ((lane >> 2) & 3);
    const unsigned wlane = (unsigned)(wrow * 64 + (((lane & 3) ^ swz64(lane >> 2)) << 4));
    const unsigned char* wsrc[SW];          // wave-uniform source of piece s at the next k-step to issue
    const unsigned char* wnext[SW];         // the same at k-step 0 of the workgroup's next tile
    int wdst[SW];                           // its byte offset inside a ring slot
    auto weight_base = [&](const KArgs a, int s, int t_n0) -> const unsigned char* {
        int idx = wave + 8 * s;
        if (idx >= 2 * WN) idx -= 8;
        const int ic = idx >= 0 ? idx : 0;                       // (WN < 4: waves >= 2 WN have no piece at all and send piece 0)
        const int icc = ic < 2 * WN ? ic : 0;
        const int plane = icc >= WN ? 1 : 0;
        const int j = icc - plane * WN;
        int rg = t_n0 + 16 * j;
        if (rg >= a->wrows) rg = a->wrows - 16;                  // row groups past the packed rows: columns never stored
        return reinterpret_cast<const unsigned char*>(plane ? a->w_lo : a->w_hi) + (long long)rg * 64;
    };
#pragma unroll
    for (int s = 0; s < SW; ++s) {
        int idx = wave + 8 * s;
        if (idx >= 2 * WN) idx -= 8;
        const int ic = idx >= 0 ? idx : 0;
        const int icc = ic < 2 * WN ? ic : 0;
        const int plane = icc >= WN ? 1 : 0;
        const int j = icc - plane * WN;
        wsrc[s] = weight_base(a0, s, n0);
        wnext[s] = weight_base(a0, s, nn0);
        wdst[s] = (plane * BN + 16 * j) * 64;
    }
    long long step_bytes, chunk_bytes;
    const unsigned char* in_hi0;            // plane bases of a tile's first chunk
    const unsigned char* in_lo0;
    {
        const KArgs a = fresh();
        step_bytes = (long long)a->wrows * 64;
        chunk_bytes = a->in_rows * 64;
        in_hi0 = reinterpret_cast<const unsigned char*>(a->in_hi);
        in_lo0 = reinterpret_cast<const unsigned char*>(a->in_lo);
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
            pin_s(wsrc[s]);     // (pinned: left to itself hipcc sinks this bookkeeping into the MFMA phase, a clump of ~25 scalar instructions)
        }
        kleft = more ? kleft - 1 : (has_next ? nk - 1 : 0);
        wr_off = wr_off + WSLOT == NB * WSLOT ? 0 : wr_off + WSLOT;
        pin_s(kleft);
        pin_s(wr_off);
    };
    const unsigned char* hsrc_hi = in_hi0;      // plane bases of the chunk whose halo is issued next
    const unsigned char* hsrc_lo = in_lo0;
    int hbuf = 0;                         // halo buffer (byte offset) that chunk goes to
    auto issue_halo = [&](auto sc) {      // halo piece wave + 8 S (waves 2..7, S = 5: piece wave + 32 again)
        constexpr int S = decltype(sc)::value;
        int k = wave + 8 * S;
        if (S == 5 && k >= 2 * HALO_PLANE_PIECES) k -= 8;
        dma16((k >= HALO_PLANE_PIECES ? hsrc_lo : hsrc_hi) + hoff[S], smem + hbuf + k * 1024);
    };
    int chunks_left = nchunks - 1;        // chunks of the issuer's tile after the one whose halo is issued next
    auto halo_advance = [&]() {
        if (chunks_left > 0) {
            hsrc_hi += chunk_bytes;
            hsrc_lo += chunk_bytes;
            --chunks_left;
        } else if (has_next) {
            hsrc_hi = in_hi0;
            hsrc_lo = in_lo0;
            if (!halo_inside(nox0, noy0)) setup_halo(nimg, nox0, noy0);     // (a tile on the image border: the general form; else done behind the chunk's last MFMAs)
            chunks_left = nchunks - 1;
        }
        hbuf = HALO_BYTES - hbuf;
    };

    f32x4 acc[2][WN], pend[2][WN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};        // (the first tile "finishes" an empty one: its stores are all masked)

    // tail k-steps: lane group g reads slot 0 of the halo pixel of tap 4t + g (taps 9..11 meet zero weights: tap 8 again)
    int dtail = 0;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int tap = (4 * t + g) < 9 ? 4 * t + g : 8;
        const int ty = tap / 3;
        dtail |= (ty * HW_ + (tap - 3 * ty)) << (8 * t);
    }

    // ---- prologue: halo of chunk 0, weights of k-steps 0 .. LA-1 (a tile's epilogue constants go out at the top of the tile loop: the
    // counted waits of a tile's first k-steps count that piece) ----
    static_for<0, 6>([&](auto sc) { issue_halo(sc); });
    halo_advance();
#pragma unroll
    for (int u = 0; u < LA; ++u) issue_weights();
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();          // the second group runs one phase behind

    f16x8 xh[2], xl[2], xs[2], wh[WN], wl[WN];
    const unsigned wfrag = ring0 + (unsigned)(r * 64 + ((g ^ swz64(r)) << 4));
    const int prow = 2 * wave * HW_ + r;
    unsigned xa[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) xa[k] = halo0 + (unsigned)(prow * 64 + ((g ^ swz64(prow + k)) << 4));
    int rd_off = 0;
    int hcur = 0;
    int seq = 0;
    int dxa = 0;                    // (uniform) the chunk's last k-step: byte step of the fragment bases to the other halo buffer,
    bool hsw = false;               // does the halo stream move to the next tile with this chunk (and is that tile away from the border),
    unsigned hsw_tb = 0u;           // and that tile's halo base
    const float* cptr_n = nullptr;  // the next tile's constants source (per lane)
#ifdef ATMVFI_STAMP
    unsigned stv0 = 0u, stv1 = 0u;
#endif

    // ---- the PENDING tile: its sums (pend), the constants buffer with its bias / slopes, and -- per lane -- where its outputs go and
    // which of them exist, as ONE mask register: bit 2 jp + i = n-tile pair jp, pixel row i: the lane's 8-channel plane store
    const float* pcst = cst_base;
    const int cb = 8 * (g & 1) + 4 * (g >> 1);
    constexpr int NP = (WN + 1) / 2;
    unsigned pm = 0u;
    unsigned po1 = 0u;              // byte offset of (pixel row 0, the lane's first channel of pair 0) in the sink's planes
    // store bases and strides: the scalars the units read (loaded once; the k-loop leaves room for them)
    void *o_hi, *o_lo;
    unsigned row_p, pair_1;
    const float *c_bias, *c_prelu;      // ... and what the tile's constants piece and the store placement read
    int c_cout, c_c0;
    {
        const KArgs a = fresh();
        c_bias = a->bias; c_prelu = a->prelu; c_cout = a->Cout; c_c0 = a->out_c0;
        o_hi = a->out_hi; o_lo = a->out_lo;
        row_p = (unsigned)a->W * 64u;                           // one pixel row down: bytes in a plane
        pair_1 = (unsigned)a->plane_rows * 64u;                 // one 32-channel chunk (= n-tile pair) on: bytes
    }
    // (computed for the CURRENT tile behind MFMAs of its first chunk -- pp_a .. pp_c, k-step 6 -- into pm_n / po_n, which become pm / po1
    // when the tile ends: at the tile boundary the SIMD partner is in an MFMA phase at raised priority, and vector instructions issued there
    // cost ~20 cycles each)
    unsigned pm_n = 0u, po_n = 0u, pp_row = 0u;
    bool pp_l0 = false, pp_l1 = false;
    auto pp_a = [&](int simg, int sox0, int soy0) {
        const int oy = soy0 + 2 * wave, ox = sox0 + r;
        pp_l0 = oy < H && ox < W;
        pp_l1 = oy + 1 < H && ox < W;
        pp_row = (unsigned)((simg * H + oy) * W + ox);        // (garbage for dead lanes: they store nothing)
    };
    auto pp_b = [&](int sn0) {
        const int climit = (c_cout + 7) & ~7;
        unsigned m = 0u;
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) {
            const int jt = g < 2 ? 2 * jp : 2 * jp + 1;
            const bool ok = jt < WN && sn0 + 16 * jt + 8 * (g & 1) < climit;
            m |= (ok && pp_l0 ? 1u : 0u) << (2 * jp);
            m |= (ok && pp_l1 ? 1u : 0u) << (2 * jp + 1);
        }
        pm_n = m;
    };
    auto pp_c = [&](int sn0) {
        const int cs = sn0 + (g < 2 ? 0 : 16) + 8 * (g & 1);          // first of the 8 channels this lane stores for n-tile pair 0
        const int c1 = c_c0 + cs;
        po_n = (unsigned)(c1 >> 5) * pair_1 + pp_row * 64u + (unsigned)(c1 & 31) * 2u;
    };
    // the tile's constants piece: its per-lane source, computed ahead (k-step 7 of the tile in front), and the one DMA instruction
    const float* cptr = nullptr;
    auto consts_ptr = [&](int t_n0) -> const float* {
        constexpr int NPIECE = (3 * BN + 63) / 64;
        const int piece = wave % NPIECE;
        const int t = piece * 64 + lane;
        const int row = t / BN;
        const int col = t_n0 + t - row * BN;
        const float* src = row == 0 ? c_bias : row == 1 ? c_prelu : nullptr;
        return (src && row < 3 && col < c_cout) ? src + col : &kEpilogueDefaults[row == 0 ? 0 : 1];
    };
    auto consts_issue = [&](const float* pl, float* cst) {
        constexpr int NPIECE = (3 * BN + 63) / 64;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pl,
                                         (__attribute__((address_space(3))) void*)(cst + (wave % NPIECE) * 64), 4, 0, 0);
    };

    // ---- the atoms of unit U = (n-tile pair jp, pixel row i), IN PLACE on the two vectors src[i][2 jp], src[i][2 jp + 1] (an odd WN's
    // last pair: the one n-tile twice, the upper half-wave's store is masked).  State between atoms: registers.
    constexpr int NUNIT = 2 * NP;
    constexpr int NATOM = DE_NATOMS;
    f32x4 tc, ts;                   // constants in flight / slope x value
    unsigned hh[4], ll[4];          // split halves: dwords 0, 1 of n-tile j0, 2, 3 of n-tile j1
    unsigned sp_h, sp_l;            // a split between its parts: hi pair, lo' under construction, values x 1024
    float sp_x, sp_y;
    unsigned st_m, st_o;            // a store's mask word and byte offset
    auto unit_atom = [&](auto qc, f32x4 (&src)[2][WN], const float* scst) {
        constexpr int Q = decltype(qc)::value;
        constexpr int U = Q / NATOM, A = de_atom(Q % NATOM);
        constexpr int jp = U >> 1, i = U & 1;
        constexpr int j0 = 2 * jp, j1 = (2 * jp + 1 < WN) ? 2 * jp + 1 : 2 * jp;
        constexpr bool two = 2 * jp + 1 < WN;
        auto ld = [&](f32x4& d, int off) { d = *reinterpret_cast<const f32x4*>(scst + off + cb); };
        if constexpr (A >= 10 && A < 30) {
            constexpr int t = (A - 10) / 10, k = (A - 10) % 10;
            constexpr int j = t ? j1 : j0;
            if constexpr (t == 0 || two) {
                f32x4& P = src[i][j];
                if constexpr (k == 0) { ld(tc, 16 * j); ld(ts, BN + 16 * j); }
                else if constexpr (k == 1) { P.x = P.x + tc.x; P.y = P.y + tc.y; }
                else if constexpr (k == 2) { P.z = P.z + tc.z; P.w = P.w + tc.w; }
                else if constexpr (k == 3) { tc.x = ts.x * P.x; tc.y = ts.y * P.y; }
                else if constexpr (k == 4) { tc.z = ts.z * P.z; tc.w = ts.w * P.w; }
                else if constexpr (k == 5) P.x = sel_neg(P.x, tc.x);
                else if constexpr (k == 6) P.y = sel_neg(P.y, tc.y);
                else if constexpr (k == 7) P.z = sel_neg(P.z, tc.z);
                else if constexpr (k == 8) P.w = sel_neg(P.w, tc.w);
            }
        } else {
            constexpr int k = A - 60;
            if constexpr (k < 12) {
                constexpr int pr = k / 3, part = k % 3;                   // value pair: 0, 1 of n-tile j0, 2, 3 of n-tile j1
                f32x4& P = src[i][pr < 2 ? j0 : j1];
                if constexpr (part == 0) split_a(pr & 1 ? P.z : P.x, pr & 1 ? P.w : P.y, sp_h, sp_x);
                else if constexpr (part == 1) split_b(pr & 1 ? P.w : P.y, sp_h, sp_x, sp_y, sp_l);
                else { split_c(sp_h, sp_y, sp_l); hh[pr] = sp_h; ll[pr] = sp_l; }
            } else if constexpr (k == 12 || k == 13) {
                unsigned (&v)[4] = k == 12 ? hh : ll;
                const auto s0 = __builtin_amdgcn_permlane32_swap(v[0], v[2], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(v[1], v[3], false, false);
                v[0] = s0[0]; v[2] = s0[1]; v[1] = s1[0]; v[3] = s1[1];
            } else if constexpr (k == 14) {
                st_m = pm & (1u << (2 * jp + i));
                st_o = po1 + ((i ? row_p : 0u) + (unsigned)jp * pair_1);
            } else {
                const u32x4 hv = {hh[0], hh[1], hh[2], hh[3]}, lv = {ll[0], ll[1], ll[2], ll[3]};
                mstore_pair16(st_m, st_o, hv, lv, o_hi, o_lo);
            }
        }
    };
    constexpr int QTOT = NUNIT * NATOM;           // the atoms of a tile: de_slot / de_q0 deal them over the next tile's MFMA slots
    static_assert(QTOT == de_qtot(WN), "unit count");
    constexpr int MS = 6 * WN;                    // MFMAs of a k-step

    const f16x8 k2m10 = {(_Float16)0.0009765625f, (_Float16)0.0009765625f, (_Float16)0.0009765625f, (_Float16)0.0009765625f,
                         (_Float16)0.0009765625f, (_Float16)0.0009765625f, (_Float16)0.0009765625f, (_Float16)0.0009765625f};
    // One k-step of one wave.  T = tap (regular chunk) or tail step; FIRST: the tile's first chunk, whose MFMA phases carry the copy and
    // the pending tile's epilogue.
    auto kstep = [&](auto tc_, auto tailc, auto firstc) {
        constexpr int T = decltype(tc_)::value;
        constexpr bool TAIL = decltype(tailc)::value;
        constexpr bool FIRST = decltype(firstc)::value;
        [[maybe_unused]] constexpr int SIDX = 4 * (FIRST ? T : TAIL ? 18 + T : 9 + T);
        DSTAMP(SIDX)
        // ---------------- read phase ----------------
        if constexpr (!TAIL) {
            constexpr int C0 = (T / 3) * HW_ + T % 3, C1 = C0 + HW_;
            lds_read16<64 * C0>(xh[0], xa[C0 & 7]);
            lds_read16<64 * C0 + HALO_LO>(xl[0], xa[C0 & 7]);
            lds_read16<64 * C1>(xh[1], xa[C1 & 7]);
            lds_read16<64 * C1 + HALO_LO>(xl[1], xa[C1 & 7]);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int p = prow + i * HW_ + ((dtail >> (8 * T)) & 0xff);
                const unsigned addr = halo0 + (unsigned)hcur + (unsigned)(p * 64 + (swz64(p) << 4));
                lds_read16<0>(xh[i], addr);
                lds_read16<HALO_LO>(xl[i], addr);
            }
        }
        const unsigned wa = wfrag + (unsigned)rd_off;
        static_for<0, WN>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            lds_read16<j * 1024>(wh[j], wa);
            lds_read16<j * 1024 + BN * 64>(wl[j], wa);
        });
        rd_off = rd_off + WSLOT == NB * WSLOT ? 0 : rd_off + WSLOT;
        if constexpr ((!TAIL && T == 8) || (TAIL && T == 2)) {
            dxa = hcur ? -HALO_BYTES : HALO_BYTES;
            hsw = chunks_left == 0 && has_next && halo_inside(nox0, noy0);
            hsw_tb = halo_base(nimg, nox0, noy0);
        }
        issue_weights();
        if constexpr (!TAIL && T < 6) issue_halo(std::integral_constant<int, T>{});
        if constexpr (TAIL && T < 2) {
            issue_halo(std::integral_constant<int, 3 * T>{});
            issue_halo(std::integral_constant<int, 3 * T + 1>{});
            issue_halo(std::integral_constant<int, 3 * T + 2>{});
        }
        // Own pieces of the next k-step landed.  Behind its weights (issued LA - 1 read phases ago) went SW weight pieces per phase, the
        // halo pieces of the phases T + 1 - LA .. T of this chunk and -- first LA - 1 k-steps of a tile -- the tile's constants piece.
        // What the previous tile's last phases issued besides their weights (the three-piece halo issues of tail steps) is not counted
        // (counting fewer operations than are behind only waits longer), nor are the rare fp32 stores -- but the plane-sink stores of the
        // epilogue units ARE: vmcnt counts loads, stores and LDS-DMA together, in issue order (MI355X_MICROARCH.md; hipcc's own counted
        // waits rely on it), so a wait that does not count them also waits for stores issued half a k-step earlier.
        {
            constexpr int lo = T + 1 - LA > 0 ? T + 1 - LA : 0;
            constexpr int hi = T < 5 ? T : 5;
            constexpr int halos = !TAIL ? (hi >= lo ? hi - lo + 1 : 0) : (lo <= 0 ? 3 : 0) + ((lo <= 1 && T >= 1) ? 3 : 0);
            constexpr int consts = (FIRST && T < LA - 1) ? 1 : 0;
            // The LAST k-step of a chunk is also where the next chunk's halo has to be complete (the next read phase starts with it): only
            // what was issued after its last piece may stay in flight -- the weights of taps 6, 7, 8 behind a regular chunk's tap-5 piece
            // (at least LA - 1 phases of them), the weights of tail step 2 behind the three pieces of tail step 1.
            constexpr int stores = (FIRST && (ATMVFI_DE_EXP & 3) == 0) ? de_stores_behind(WN, T, LA) : 0;
            constexpr int N = ((!TAIL && T == 8) ? (LA - 1) * SW : (TAIL && T == 2) ? SW : (LA - 1) * SW + halos + consts) + stores;
            static_assert(N < 64, "vmcnt is a 6-bit counter");
            static_assert(LA == 4, "de_stores_behind assumes a lookahead of 4 k-steps");
            wait_vm<N>();
        }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(xh[0]), "+v"(xh[1]), "+v"(xl[0]), "+v"(xl[1]));
#pragma unroll
        for (int j = 0; j < WN; ++j) asm volatile("" : "+v"(wh[j]), "+v"(wl[j]));
        // lo' planes carry (x - hi) * 1024: scaled back here, in fp16, so that all three products of a k-step go into ONE accumulator
        xl[0] = xl[0] * k2m10;
        xl[1] = xl[1] * k2m10;
        xs[0] = xh[0] * k2m10;      // hi * 2^-10 for the third pass' lo'(weights) x hi(activations) product (the weights' lo' stays scaled)
        xs[1] = xh[1] * k2m10;
        DSTAMP(SIDX + 1)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        DSTAMP(SIDX + 2)
        // ---------------- MFMA phase ----------------
        // three passes of 2 WN MFMAs (hi x hi, hi x lo * 2^-10, lo' x hi * 2^-10); MFMA number m of the k-step:
        if constexpr ((ATMVFI_DE_EXP & 4) == 0) __builtin_amdgcn_s_setprio(1);
        static_for<0, MS>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            constexpr int pass = m / (2 * WN), j = (m % (2 * WN)) >> 1, i = m & 1;
            if constexpr (FIRST && T == 0 && pass == 0) {
                // a new tile begins: the finished sums move to pend, the new ones start from C = 0
                pend[i][j] = acc[i][j];
                asm volatile("" : "+v"(pend[i][j]));
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xh[i], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            } else if constexpr (pass == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xh[i], acc[i][j], 0, 0, 0);
            else if constexpr (pass == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], xl[i], acc[i][j], 0, 0, 0);
            else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[j], xs[i], acc[i][j], 0, 0, 0);
            constexpr bool LASTK = (!TAIL && T == 8) || (TAIL && T == 2);       // the chunk's last k-step
            if constexpr (FIRST && T == 6) {
                // where this tile's outputs go (needed when the NEXT tile runs its epilogue), in three pieces
                if constexpr (m == 0) pp_a(img, ox0, oy0);
                else if constexpr (m == MS / 3) pp_b(n0);
                else if constexpr (m == 2 * MS / 3) pp_c(n0);
            }
            if constexpr (FIRST && T == 7 && m == 0) cptr_n = consts_ptr(nn0);
            if constexpr (LASTK) {
                // on to the other halo buffer: the eight fragment bases, one or two behind an MFMA; and, if the DMA stream moves to the
                // workgroup's next tile with this chunk, that tile's six halo offsets
                static_for<m * 8 / MS, (m + 1) * 8 / MS>([&](auto kc) { xa[decltype(kc)::value] += (unsigned)dxa; });
                if constexpr (m == MS - 1) {
                    if (hsw) {
#pragma unroll
                        for (int s2 = 0; s2 < 6; ++s2) hoff[s2] = hsw_tb + hrel[s2];
                    }
                }
            }
            if constexpr (FIRST && (ATMVFI_DE_EXP & 1) == 0) {
                constexpr int gs = de_slot(WN, T, m);       // slot number among the carrying slots
                if constexpr (gs >= 0) {
                    constexpr int qa = de_q0(WN, gs), qb = de_q0(WN, gs + 1);      // atoms q with floor(q NSLOT / QTOT) == gs
                    static_for<qa, (qb < QTOT ? qb : QTOT)>([&](auto qc) { unit_atom(qc, pend, pcst); });
                }
            }
            if constexpr (FIRST || LASTK) __builtin_amdgcn_sched_barrier(0);
        });
        __builtin_amdgcn_s_setprio(0);
        DSTAMP(SIDX + 3)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto next_chunk = [&]() {          // (scalar work only: the fragment bases moved behind the last k-step's MFMAs)
        hcur += dxa;
        halo_advance();
    };

    cptr = consts_ptr(n0);
    for (;;) {
        float* cst = cst_base + (seq & 1) * CSTF;
        // (this sits in the wave's read-phase slot of the tile's first k-step: the SIMD partner is in its last MFMA phase of the previous
        // tile.)  This tile's constants: needed a whole tile from now, by its deferred epilogue.
        consts_issue(cptr, cst);
        static_for<0, 9>([&](auto tc_) { kstep(tc_, std::false_type{}, std::true_type{}); });
        next_chunk();
        for (int c = 1; c < nfull; ++c) {
            static_for<0, 9>([&](auto tc_) { kstep(tc_, std::false_type{}, std::false_type{}); });
            next_chunk();
        }
        if (ktail) {
            static_for<0, 3>([&](auto tc_) { kstep(tc_, std::true_type{}, std::false_type{}); });
            next_chunk();
        }
        if (!has_next) break;
        // on to the next tile without a pause: its first halo and its first LA k-steps of weights are in LDS or on their way
        const KArgs a = fresh();
        pm = pm_n;
        po1 = po_n;
        cptr = cptr_n;
        pcst = cst;
        vb += grid;
        img = nimg; ox0 = nox0; oy0 = noy0; n0 = nn0;
        ++seq;
        has_next = decode(a, vb + grid, nimg, nox0, noy0, nn0);
#pragma unroll
        for (int s = 0; s < SW; ++s) wnext[s] = weight_base(a, s, nn0);
    }
    // ---- the last tile's epilogue, in the open: the first group waits for the second one's last MFMA phase, everything in flight (the
    // tile's constants among it) lands, then the units one after the other, straight from the accumulators
    if (grp == 0) __builtin_amdgcn_s_barrier();
    wait_vm<0>();
    {
        const float* cst = cst_base + (seq & 1) * CSTF;
        pp_a(img, ox0, oy0);
        pp_b(n0);
        pp_c(n0);
        pm = pm_n;
        po1 = po_n;
        static_for<0, QTOT>([&](auto qc) { unit_atom(qc, acc, cst); });
    }
#ifdef ATMVFI_STAMP
    {
        const KArgs a = fresh();
        if (a->stamp) {
            unsigned* o = reinterpret_cast<unsigned*>(a->stamp) + ((long long)blockIdx.x * 8 + wave) * 128;
            o[lane] = stv0;
            o[64 + lane] = stv1;
        }
    }
#endif
}

template <int WN>
int launch_de_instance(const Conv3PDev& ds, int grid, size_t lds, hipStream_t s) {
    const hipError_t e = atmvfi::allow_dynamic_lds<conv3x3_planes_de_kernel<WN>>(lds);
    ATMVFI_REQUIRE(e == hipSuccess, ATMVFI_ELAUNCH, "conv3x3_planes: hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(conv3x3_planes_de_kernel<WN>, dim3((unsigned)grid), dim3(512), lds, s, ds);
    return atmvfi::check_launch("conv3x3_planes (deferred epilogue)");
}

}  // namespace

namespace atmvfi {
int launch_planes_de(int wn, const Conv3PDev& ds, int grid, size_t lds, hipStream_t s) {
    switch (wn) {
        case 1: return launch_de_instance<1>(ds, grid, lds, s);
        case 2: return launch_de_instance<2>(ds, grid, lds, s);
        case 3: return launch_de_instance<3>(ds, grid, lds, s);
        case 4: return launch_de_instance<4>(ds, grid, lds, s);
        case 5: return launch_de_instance<5>(ds, grid, lds, s);
        case 6: return launch_de_instance<6>(ds, grid, lds, s);
        default: return launch_de_instance<7>(ds, grid, lds, s);
    }
}
}  // namespace atmvfi

// Shared descriptor and epilogue of the two GEMM engines (exact-fp32 MFMA: gemm_conv.hip;
// split-precision f16x3: gemm_f16x3.hip).
#pragma once
#include "common.h"

namespace atmvfi {

struct GemmDev {
    int mode;
    const float* in;
    int in_ld, H, W, Cin;
    long long in_gstride;
    int in_rpg;
    const float* weight;
    int wrows;        // weight rows present (multiple of 16)
    int ktot;         // floats per weight row = taps * cin_pad
    int cin_pad;      // Cin rounded up to 16
    int cpt;          // chunks per tap = cin_pad / 16
    int nchunks;      // taps * cpt
    int Cout;         // real output channels (DECONV: per-position channels)
    int coutp;        // DECONV: Cout rounded up to 4
    int kw, stride, pad, dil;
    int Ho, Wo;
    long long M;
    float* out;
    int out_ld;
    long long out_gstride;
    int out_rpg;
    const int* out_row_map;
    const float* bias;
    const float* prelu;
    const float* in_prelu;
    const float* residual;
    int res_ld;
    // split-precision operands (f16x3 engine only)
    const _Float16* w_hi;
    const _Float16* w_lo;
    int cin_pad32;    // Cin rounded up to 32
    int cpt32;        // 32-channel chunks per tap
    int nchunks32;    // taps * cpt32
    int ktot32;       // halves per split weight row = taps * cin_pad32
    const _Float16* a_hi;   // split-plane activations (gemm_split.hip), else null
    const _Float16* a_lo;
    const _Float16* a_hi2;  // CONV mode: input chunks >= split_chunks come from a second plane buffer with in_ld2 rows per chunk
    const _Float16* a_lo2;
    int in_ld2, split_chunks;
    int in_N;               // images (CONV with planes: the zero row is row N*H*W)
    unsigned long long* stamp;   // diagnostic builds only (ATMVFI_STAMP)
    int dbg;          // diagnostic ablations (gemm_split.hip, env ATMVFI_SPLIT_DEBUG): 1 = no stores, 2 = one k-step
    int mchunk;       // row tiles per XCD = ceil(row tiles / 8)
    int vblocks;      // virtual blocks (tiles incl. XCD padding) walked by the persistent grid
    int nblocks;      // column blocks per row tile (set by the launcher; XCD-aware tile order)
    // optional plane sink of the f16x3 engines: the result again (or only: `out` may then be null) as split planes, chunk major
    // (common.h RowSink), channel co of output row r at plane row / channel (r % out_rpg, out_c0 + (r / out_rpg) * out_gc + co)
    // (out_rpg == 0: (r, out_c0 + co)); DECONV: r = output pixel
    _Float16* out_hi;
    _Float16* out_lo;
    long long out_plane_rows;
    int out_c0, out_gc;
    int force_wn;     // per-call tile-width override of gemm_f16x3 (0 = cost model)
    // split-K of under-filled long-K launches (gemm_duo.hip, round 4): gridDim.y = ksplit workgroups per tile, each over a contiguous
    // range of k-steps, raw fp32 partial sums to part + split * part_stride (rows of part_ld floats); gemm_splitk_reduce_kernel adds
    // them in split order and runs the epilogue.  0 / 1 = off.
    int ksplit;
    float* part;
    long long part_stride;
    int part_ld;
    int fit32;        // gemm_pp: rows x out_ld x 4 and rows x res_ld x 4 bytes fit 32 bits (unmapped, ungrouped fp32 rows: 32-bit store offsets)
    int pfit32;       // gemm_pp: a plane of the sink (plane rows x 64 bytes) fits 32 bits: 32-bit row offsets from per-lane chunk pointers
};


// Output row of GEMM row m: NHWC pixel (CONV), scattered/grouped token row (LINEAR) or the (0,0)
// position of the 2x2 output patch (DECONV).  Returns false when the row is dropped by the map.
// prow / pc0: row and first channel of the row in the optional plane sink.
__device__ __forceinline__ bool gemm_out_row(const GemmDev& a, long long m, float*& orow, const float*& rrow, long long& prow, int& pc0) {
    pc0 = a.out_c0;
    if (a.mode == ATMVFI_GEMM_DECONV) {
        const int hw = a.H * a.W;
        // 32-bit division whenever the row index allows (always, for this network): the 64-bit one is ~70 instructions per row
        const int n = (m >> 31) ? (int)(m / hw) : (int)((unsigned)m / (unsigned)hw);
        const int rem = (int)(m - (long long)n * hw);
        const int y = rem / a.W;
        const int x = rem - y * a.W;
        prow = ((long long)n * a.Ho + 2 * y) * a.Wo + 2 * x;
        orow = a.out + prow * a.out_ld;
    } else {
        long long ro = m;
        if (a.out_row_map) ro = a.out_row_map[m];
        if (ro < 0) return false;
        prow = ro;
        long long off = ro * (long long)a.out_ld;
        if (a.out_rpg > 0) {
            const long long grp = ro / a.out_rpg;
            prow = ro - grp * a.out_rpg;
            off = grp * a.out_gstride + prow * (long long)a.out_ld;
            pc0 += (int)grp * a.out_gc;
        }
        orow = a.out + off;
    }
    rrow = a.residual ? a.residual + m * (long long)a.res_ld : nullptr;
    return true;
}
// The same with the row-map entry already fetched (`ro`: mapped row, or m when there is no map; >= 0): lets a kernel batch the map
// loads of several rows and rebuild the pointers only when it stores (gemm_split.hip's epilogue keeps one int per row).
__device__ __forceinline__ void gemm_out_row_at(const GemmDev& a, long long m, long long ro, float*& orow, long long& prow, int& pc0) {
    pc0 = a.out_c0;
    if (a.mode == ATMVFI_GEMM_DECONV) {
        const int hw = a.H * a.W;
        const int n = (m >> 31) ? (int)(m / hw) : (int)((unsigned)m / (unsigned)hw);
        const int rem = (int)(m - (long long)n * hw);
        const int y = rem / a.W;
        const int x = rem - y * a.W;
        prow = ((long long)n * a.Ho + 2 * y) * a.Wo + 2 * x;
        orow = a.out + prow * a.out_ld;
    } else {
        prow = ro;
        long long off = ro * (long long)a.out_ld;
        if (a.out_rpg > 0) {
            const long long grp = ro / a.out_rpg;
            prow = ro - grp * a.out_rpg;
            off = grp * a.out_gstride + prow * (long long)a.out_ld;
            pc0 += (int)grp * a.out_gc;
        }
        orow = a.out + off;
    }
}
__device__ __forceinline__ bool gemm_out_row(const GemmDev& a, long long m, float*& orow, const float*& rrow) {
    long long prow;
    int pc0;
    return gemm_out_row(a, m, orow, rrow, prow, pc0);
}

// Per-n-tile channel constants (bias, PReLU slope) of GEMM columns nb..nb+3, loaded ONCE per tile as
// 16-byte vectors (the host guarantees 16-byte aligned bias/prelu/residual; co is a multiple of 4).
// Scalar per-element loads here cost 100+ dependent dword loads per lane per tile and dominated
// the short-K layers.  Missing bias -> 0, missing PReLU -> slope 1 (identity), so the math is branch-free.
struct ChanVec {
    f32x4 b, p;
    int co;        // first output channel (DECONV: within the 2x2 position), -1 = nothing to store
    int q;         // DECONV: position index a*2+b
    int nvalid;
};

__device__ __forceinline__ ChanVec gemm_chan_vec(const GemmDev& a, int nb) {
    ChanVec c;
    c.q = 0;
    c.co = nb;
    if (a.mode == ATMVFI_GEMM_DECONV) {
        c.q = (nb >= a.coutp) + (nb >= 2 * a.coutp) + (nb >= 3 * a.coutp);
        c.co = nb - c.q * a.coutp;
        if (nb >= 4 * a.coutp) c.co = -1;
    }
    if (c.co >= a.Cout) c.co = -1;
    c.nvalid = c.co < 0 ? 0 : a.Cout - c.co;
    c.b = (f32x4){0.f, 0.f, 0.f, 0.f};
    c.p = (f32x4){1.f, 1.f, 1.f, 1.f};
    if (c.nvalid >= 4) {
        if (a.bias) c.b = *reinterpret_cast<const f32x4*>(a.bias + c.co);
        if (a.prelu) c.p = *reinterpret_cast<const f32x4*>(a.prelu + c.co);
    } else if (c.nvalid > 0) {
        float bb[4] = {0.f, 0.f, 0.f, 0.f}, pp[4] = {1.f, 1.f, 1.f, 1.f};
        for (int e = 0; e < c.nvalid; ++e) {
            if (a.bias) bb[e] = a.bias[c.co + e];
            if (a.prelu) pp[e] = a.prelu[c.co + e];
        }
        c.b = (f32x4){bb[0], bb[1], bb[2], bb[3]};
        c.p = (f32x4){pp[0], pp[1], pp[2], pp[3]};
    }
    return c;
}

// Store four consecutive GEMM columns of one row: + bias, PReLU, + residual, one 16-byte store.
__device__ __forceinline__ void gemm_store4(const GemmDev& a, float* orow, const float* rrow, const ChanVec& c, f32x4 v) {
    if (c.co < 0) return;
    float* optr = orow;
    if (a.mode == ATMVFI_GEMM_DECONV) optr = orow + ((long long)(c.q >> 1) * a.Wo + (c.q & 1)) * a.out_ld;
    v += c.b;
    v.x = v.x > 0.f ? v.x : c.p.x * v.x;
    v.y = v.y > 0.f ? v.y : c.p.y * v.y;
    v.z = v.z > 0.f ? v.z : c.p.z * v.z;
    v.w = v.w > 0.f ? v.w : c.p.w * v.w;
    if (c.nvalid >= 4) {
        if (rrow) v += *reinterpret_cast<const f32x4*>(rrow + c.co);
        *reinterpret_cast<f32x4*>(optr + c.co) = v;
    } else {
        const float vv[4] = {v.x, v.y, v.z, v.w};
        for (int e = 0; e < c.nvalid; ++e) optr[c.co + e] = vv[e] + (rrow ? rrow[c.co + e] : 0.f);
    }
}

// ---- epilogue constants through LDS -------------------------------------------------------------------------------
// With the constants fetched per n-tile from global memory inside the epilogue (above), hipcc serialises the epilogue into
// load -> s_waitcnt vmcnt(0) -> store -> s_waitcnt vmcnt(0) round trips (the per-lane tail branches defeat its counting):
// ~48 dependent L2 round trips = 14 k of the 60 k ticks of a K = 384 tile.  The f16x3 engines therefore fetch the 2*BN
// constants of their column block ONCE, by LDS-DMA (no VGPR, no early wait), as cst[c] = bias, cst[BN + c] = PReLU slope of
// GEMM column n0 + c (0 / 1 when absent or out of range) and read them back with one ds_read_b128 each: the epilogue has
// no global load left (except an optional residual: one batch of loads and one wait per output row).
template <int BN>
__device__ __forceinline__ void gemm_dma_consts(const GemmDev& a, int n0, float* cst, int wave, int lane) {
    dma_epilogue_consts<BN>(a.bias, a.prelu, n0, cst, wave, lane, [&](int col) {
        int co = col;
        if (a.mode == ATMVFI_GEMM_DECONV) {
            const int q = col / a.coutp;
            co = q < 4 ? col - q * a.coutp : a.Cout;
        }
        return co < a.Cout ? co : -1;
    });
}
constexpr int gemm_const_floats(int BN) { return epilogue_const_floats(BN); }

struct ChanPos {
    int co;        // first output channel (DECONV: within the 2x2 position)
    int q;         // DECONV: position index a*2+b
    int nvalid;    // channels to store (<= 0: none)
};
__device__ __forceinline__ ChanPos gemm_chan_pos(const GemmDev& a, int nb) {
    ChanPos c;
    c.q = 0;
    c.co = nb;
    if (a.mode == ATMVFI_GEMM_DECONV) {
        c.q = (nb >= a.coutp) + (nb >= 2 * a.coutp) + (nb >= 3 * a.coutp);
        c.co = nb - c.q * a.coutp;
        if (nb >= 4 * a.coutp) c.co = a.Cout;
    }
    c.nvalid = a.Cout - c.co;
    return c;
}
// v: four consecutive GEMM columns of one row; b, p: their bias / slope (from LDS); res: residual values (or zero)
// prow / pc0: the row's position in the optional plane sink (gemm_out_row)
__device__ __forceinline__ void gemm_finish_store4(const GemmDev& a, float* orow, long long prow, int pc0, const ChanPos& c, f32x4 v,
                                                   const f32x4 b, const f32x4 p, const f32x4 res) {
    v += b;
    v.x = v.x > 0.f ? v.x : p.x * v.x;
    v.y = v.y > 0.f ? v.y : p.y * v.y;
    v.z = v.z > 0.f ? v.z : p.z * v.z;
    v.w = v.w > 0.f ? v.w : p.w * v.w;
    v += res;
    if (a.out) {
        float* optr = orow;
        if (a.mode == ATMVFI_GEMM_DECONV) optr = orow + ((long long)(c.q >> 1) * a.Wo + (c.q & 1)) * a.out_ld;
        if (c.nvalid >= 4) {
            *reinterpret_cast<f32x4*>(optr + c.co) = v;
        } else if (c.nvalid > 0) {
            optr[c.co] = v.x;
            if (c.nvalid > 1) optr[c.co + 1] = v.y;
            if (c.nvalid > 2) optr[c.co + 2] = v.z;
        }
    }
    if (a.out_hi && c.nvalid > 0) {
        if (c.nvalid < 4) {            // channels past Cout inside this group of 4: the planes' pad channels, written as zero
            v.y = c.nvalid > 1 ? v.y : 0.f;
            v.z = c.nvalid > 2 ? v.z : 0.f;
            v.w = 0.f;
        }
        if (a.mode == ATMVFI_GEMM_DECONV) prow += (long long)(c.q >> 1) * a.Wo + (c.q & 1);
        const RowSink sink{nullptr, 0, a.out_hi, a.out_lo, a.out_plane_rows};
        sink_store4(sink, prow, pc0 + c.co, v);
    }
}
__device__ __forceinline__ f32x4 gemm_load_residual4(const float* rrow, const ChanPos& c) {
    f32x4 r = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (c.nvalid >= 4) {
        r = *reinterpret_cast<const f32x4*>(rrow + c.co);
    } else if (c.nvalid > 0) {
        r.x = rrow[c.co];
        if (c.nvalid > 1) r.y = rrow[c.co + 1];
        if (c.nvalid > 2) r.z = rrow[c.co + 2];
    }
    return r;
}

// gemm_f16x3.hip
int launch_gemm_f16x3(const GemmDev& d, int ngemm, hipStream_t stream);
// conv_small.hip (first layer: 3 input channels, direct on the vector ALU)
bool try_launch_conv3x3_small(const GemmDev& d, int kh, int* rc, hipStream_t stream);
// gemm_split.hip (LINEAR rows read from fp16 hi/lo planes by LDS-DMA)
int launch_gemm_split(const GemmDev& d, int ngemm, hipStream_t stream);
// gemm_pp.hip (the same for LINEAR / DECONV rows, ping-pong wave groups)
int launch_gemm_pp(const GemmDev& d, int ngemm, hipStream_t stream);
// gemm_duo.hip (the same contract on 128 x 128 tiles, two workgroups per CU)
int launch_gemm_duo(const GemmDev& d, int ngemm, hipStream_t stream);

}  // namespace atmvfi

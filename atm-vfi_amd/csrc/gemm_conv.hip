// Implicit-GEMM contraction engine for gfx950: Conv2d / Linear / ConvTranspose2d(k2,s2)
// on the exact-fp32 matrix instruction v_mfma_f32_16x16x4_f32.
//
// Computes D^T = W * X^T per block:  the MFMA "A" operand is a 16-row weight fragment
// (rows = output channels), the "B" operand a 16-row activation fragment (cols = output
// pixels), so each lane ends up with 4 CONSECUTIVE output channels of ONE pixel and
// stores them as one 16-byte NHWC store with bias/PReLU/residual fused.
//
// Block = 256 threads = 4 wavefronts stacked along M; wave tile = (16*WM pixels) x (16*WN
// channels); K is walked in chunks of 16 floats = (tap, 16 input channels).  Operand tiles
// are register-staged (global -> VGPR early, VGPR -> LDS after the MFMA phase) into a
// double-buffered LDS image with one barrier per chunk.  Both tiles are [row][16] with a
// 16-byte-slot XOR swizzle so that the ds_read_b128 fragment reads and the ds_write_b128
// staging writes are bank-conflict free.  Because the k index only has to agree between
// the two operands, lane group g = lane>>4 consumes k = 4g..4g+3 of a chunk: one
// ds_read_b128 per fragment feeds four MFMA k-steps.
#include "common.h"
#include "gemm_common.h"

#include <algorithm>

namespace {

using atmvfi::GemmDev;

// 16-byte slot swizzle: slot' = slot ^ f(row), f = [0,2,3,1][(row>>2)&3].  With 64-byte rows
// this makes every ds_read_b128 lane group {16 lanes} hit 16 distinct slots of a 256-byte bank row.
__device__ __forceinline__ int swz(int row) { return (0x1320 >> (((row >> 2) & 3) << 2)) & 3; }

template <int WM, int WN>
__global__ __launch_bounds__(256) void gemm_mfma_f32(const GemmDev a) {
    fp16_saturate_on();
    constexpr int BM = 64 * WM;
    constexpr int BN = 16 * WN;
    constexpr int NB = (BN * 4 + 255) / 256;      // B float4 loads per thread
    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * 16];
    float* As = lds;                         // [2][BM][16]
    float* Bs = lds + 2 * BM * 16;           // [2][BN][16]

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int kq = t & 3;
    const long long m0 = (long long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    // ---- per-thread A-row bookkeeping (WM rows) ----
    const float* rbase[WM];
    int iy0[WM], ix0[WM];
    bool rok[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const long long m = m0 + (t >> 2) + 64 * i;
        rok[i] = m < a.M;
        const long long mm = rok[i] ? m : 0;
        if (a.mode == ATMVFI_GEMM_CONV) {
            const int hw = a.Ho * a.Wo;
            const int n = (int)(mm / hw);
            const int rem = (int)(mm - (long long)n * hw);
            const int oy = rem / a.Wo;
            const int ox = rem - oy * a.Wo;
            iy0[i] = oy * a.stride - a.pad;
            ix0[i] = ox * a.stride - a.pad;
            rbase[i] = a.in + (((long long)n * a.H + iy0[i]) * a.W + ix0[i]) * a.in_ld;
        } else {
            iy0[i] = 0;
            ix0[i] = 0;
            long long off = (a.in_rpg > 0) ? (mm / a.in_rpg) * a.in_gstride + (mm % a.in_rpg) * (long long)a.in_ld
                                           : mm * (long long)a.in_ld;
            rbase[i] = a.in + off;
        }
    }
    // ---- B-row bookkeeping ----
    const float* wbase[NB];
    bool wok[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int f = t + 256 * i;
        const int nrow = n0 + (f >> 2);
        wok[i] = (f < BN * 4) && (nrow < a.wrows);
        wbase[i] = a.weight + (long long)(wok[i] ? nrow : 0) * a.ktot + (f & 3) * 4;
    }

    f32x4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 ra[WM], rb[NB];
    int anv[WM], ac[WM];

    auto load_chunk = [&](int kc) {
        const int tap = kc / a.cpt;
        const int c0 = (kc - tap * a.cpt) * 16;
        const int ky = tap / a.kw;
        const int kx = tap - ky * a.kw;
        const int dy = ky * a.dil, dx = kx * a.dil;
        const int c = c0 + kq * 4;
        const long long toff = ((long long)dy * a.W + dx) * a.in_ld + c;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            bool ok = rok[i] && (c < a.Cin);
            if (a.mode == ATMVFI_GEMM_CONV)
                ok = ok && ((unsigned)(iy0[i] + dy) < (unsigned)a.H) && ((unsigned)(ix0[i] + dx) < (unsigned)a.W);
            // unconditional load from a clamped address + selects (a predicated load costs a branch and a vmcnt(0))
            const f32x4 v = *reinterpret_cast<const f32x4*>(ok ? rbase[i] + toff : a.in);
            anv[i] = ok ? a.Cin - c : 0;       // masking / in_prelu happen at store time: no early consumer of the loads
            ac[i] = ok ? c : 0;
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            rb[i] = *reinterpret_cast<const f32x4*>(wbase[i] + (long long)kc * 16);      // row-clamped: always valid; masked at store
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int row = (t >> 2) + 64 * i;
            f32x4 v = ra[i];
            const int nv = anv[i];
            v.x = nv > 0 ? v.x : 0.f;
            v.y = nv > 1 ? v.y : 0.f;
            v.z = nv > 2 ? v.z : 0.f;
            v.w = nv > 3 ? v.w : 0.f;
            if (a.in_prelu) {       // host pads in_prelu (uniform branch)
                const f32x4 al = *reinterpret_cast<const f32x4*>(a.in_prelu + ac[i]);
                v.x = v.x > 0.f ? v.x : al.x * v.x;
                v.y = v.y > 0.f ? v.y : al.y * v.y;
                v.z = v.z > 0.f ? v.z : al.z * v.z;
                v.w = v.w > 0.f ? v.w : al.w * v.w;
            }
            *reinterpret_cast<f32x4*>(As + ((buf * BM + row) * 16 + ((kq ^ swz(row)) << 2))) = v;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int f = t + 256 * i;
            if (f < BN * 4) {
                const int row = f >> 2;
                *reinterpret_cast<f32x4*>(Bs + ((buf * BN + row) * 16 + (((f & 3) ^ swz(row)) << 2))) =
                    wok[i] ? rb[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
    };

    const int r = lane & 15;
    const int g = lane >> 4;
    const int fslot = ((g ^ swz(r)) << 2);      // fragment rows are 16-aligned, so swz(row) == swz(r)

    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int kc = 0; kc < a.nchunks; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < a.nchunks) load_chunk(kc + 1);
        f32x4 xf[WM], wf[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i)
            xf[i] = *reinterpret_cast<const f32x4*>(As + ((buf * BM + wave * 16 * WM + 16 * i + r) * 16 + fslot));
#pragma unroll
        for (int j = 0; j < WN; ++j)
            wf[j] = *reinterpret_cast<const f32x4*>(Bs + ((buf * BN + 16 * j + r) * 16 + fslot));
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j][ks], xf[i][ks], acc[i][j], 0, 0, 0);
        if (kc + 1 < a.nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds channels nb..nb+3 of pixel m for every (i, j) ----
    float* orow[WM];
    const float* rrow[WM];
    bool live[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const long long m = m0 + wave * 16 * WM + 16 * i + r;
        live[i] = m < a.M && atmvfi::gemm_out_row(a, m, orow[i], rrow[i]);
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const atmvfi::ChanVec cv = atmvfi::gemm_chan_vec(a, n0 + 16 * j + 4 * g);
#pragma unroll
        for (int i = 0; i < WM; ++i)
            if (live[i]) atmvfi::gemm_store4(a, orow[i], rrow[i], cv, acc[i][j]);
    }
}

// -------- weight packing --------
__global__ void pack_weight_kernel(int mode, const float* __restrict__ src, float* __restrict__ dst,
                                   int Cout, int Cin, int kh, int kw, int rows, int cin_pad, int coutp) {
    fp16_saturate_on();
    const int taps = (mode == ATMVFI_GEMM_DECONV) ? 1 : kh * kw;
    const long long total = (long long)rows * taps * cin_pad;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % cin_pad);
        const int tap = (int)((idx / cin_pad) % taps);
        const int row = (int)(idx / ((long long)cin_pad * taps));
        float v = 0.f;
        if (c < Cin) {
            if (mode == ATMVFI_GEMM_DECONV) {
                const int q = row / coutp;
                const int co = row - q * coutp;
                if (q < 4 && co < Cout) v = src[(((long long)c * Cout + co) * 2 + (q >> 1)) * 2 + (q & 1)];   // IOHW
            } else if (row < Cout) {
                const int ky = tap / kw, kx = tap - ky * kw;
                v = src[(((long long)row * Cin + c) * kh + ky) * kw + kx];                                      // OIHW
            }
        }
        dst[idx] = v;
    }
}

int packed_rows(int mode, int Cout) {
    if (mode == ATMVFI_GEMM_DECONV) return atmvfi::round_up(4 * atmvfi::round_up(Cout, 4), 16);
    return atmvfi::round_up(Cout, 16);
}

template <int WM, int WN>
void launch(const GemmDev& d, int ntiles, hipStream_t s) {
    dim3 grid((unsigned)atmvfi::ceil_div64(d.M, 64 * WM), (unsigned)((ntiles + WN - 1) / WN));
    hipLaunchKernelGGL((gemm_mfma_f32<WM, WN>), grid, dim3(256), 0, s, d);
}

}  // namespace

extern "C" int64_t atmvfi_packed_weight_floats(int mode, int Cout, int Cin, int kh, int kw) {
    const int taps = (mode == ATMVFI_GEMM_DECONV) ? 1 : kh * kw;
    return (int64_t)packed_rows(mode, Cout) * taps * atmvfi::round_up(Cin, 16);
}

extern "C" int atmvfi_pack_weight(int mode, const float* src, float* dst, int Cout, int Cin, int kh, int kw, void* stream) {
    ATMVFI_REQUIRE(src && dst, ATMVFI_EINVAL, "pack_weight: null pointer");
    ATMVFI_REQUIRE(mode >= 0 && mode <= 2 && Cout > 0 && Cin > 0, ATMVFI_EINVAL, "pack_weight: bad mode/shape");
    if (mode == ATMVFI_GEMM_DECONV) ATMVFI_REQUIRE(kh == 2 && kw == 2, ATMVFI_EINVAL, "pack_weight: deconv must be 2x2");
    if (mode == ATMVFI_GEMM_LINEAR) ATMVFI_REQUIRE(kh == 1 && kw == 1, ATMVFI_EINVAL, "pack_weight: linear must be 1x1");
    const int rows = packed_rows(mode, Cout);
    const int cin_pad = atmvfi::round_up(Cin, 16);
    const int64_t total = atmvfi_packed_weight_floats(mode, Cout, Cin, kh, kw);
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mode, src, dst, Cout, Cin,
                       kh, kw, rows, cin_pad, atmvfi::round_up(Cout, 4));
    return atmvfi::check_launch("pack_weight");
}

extern "C" int atmvfi_gemm(const atmvfi_gemm_params* p, void* stream) {
    ATMVFI_REQUIRE(p, ATMVFI_EINVAL, "gemm: null params");
    const bool planes = p->in_hi || p->in_lo;
    const bool sink = p->out_hi || p->out_lo;
    ATMVFI_REQUIRE((p->in || planes) && p->weight && (p->out || sink), ATMVFI_EINVAL, "gemm: null tensor pointer");
    if (sink) {
        ATMVFI_REQUIRE(p->out_hi && p->out_lo && p->precision == ATMVFI_PREC_F16X3, ATMVFI_EINVAL,
                       "gemm: the plane sink needs both planes and precision f16x3");
        ATMVFI_REQUIRE(atmvfi::aligned16(p->out_hi) && atmvfi::aligned16(p->out_lo) && p->out_plane_c0 >= 0 && p->out_plane_c0 % 4 == 0 &&
                           p->out_plane_gc % 4 == 0, ATMVFI_EALIGN, "gemm: plane sink: 16-byte aligned planes, channel offsets multiples of 4");
        const long long orows = p->mode == ATMVFI_GEMM_DECONV ? (long long)p->N * p->Ho * p->Wo
                                : p->out_rpg > 0 ? (long long)p->out_rpg : (p->mode == ATMVFI_GEMM_CONV ? (long long)p->N * p->Ho * p->Wo : (long long)p->M);
        ATMVFI_REQUIRE(p->out_row_map || p->out_plane_rows >= orows, ATMVFI_EINVAL, "gemm: plane sink rows %lld < output rows %lld",
                       (long long)p->out_plane_rows, orows);
        ATMVFI_REQUIRE(p->out_plane_rows > 0, ATMVFI_EINVAL, "gemm: plane sink needs out_plane_rows");
    }
    if (planes) {
        ATMVFI_REQUIRE(p->in_hi && p->in_lo && p->precision == ATMVFI_PREC_F16X3 && !p->in_prelu,
                       ATMVFI_EINVAL, "gemm: split-plane input needs both planes, precision f16x3 and no in_prelu");
        ATMVFI_REQUIRE(atmvfi::aligned16(p->in_hi) && atmvfi::aligned16(p->in_lo) && p->in_rpg == 0, ATMVFI_EALIGN,
                       "gemm: split planes need 16-byte aligned pointers and plain rows");
        const long long mrows = p->mode == ATMVFI_GEMM_LINEAR ? (long long)p->M : (long long)p->N * p->H * p->W;
        // CONV reads the row behind the last pixel for taps that fall outside the image: it must exist (and be zero)
        ATMVFI_REQUIRE(p->in_ld >= mrows + (p->mode == ATMVFI_GEMM_CONV ? 1 : 0), ATMVFI_EALIGN,
                       "gemm: with split planes in_ld is the plane row count and must cover the rows (+ the zero row in CONV mode): %lld > %d",
                       mrows, p->in_ld);
        if (p->in_hi2 || p->in_lo2) {
            ATMVFI_REQUIRE(p->mode == ATMVFI_GEMM_CONV && p->in_hi2 && p->in_lo2 && atmvfi::aligned16(p->in_hi2) && atmvfi::aligned16(p->in_lo2) &&
                               p->in_ld2 >= mrows + 1 && p->in_split_chunks > 0 && p->in_split_chunks * 32 < p->Cin, ATMVFI_EINVAL,
                           "gemm: a second plane source needs CONV mode, both planes, in_ld2 > N*H*W and 0 < 32 * in_split_chunks < Cin");
        }
    }
    ATMVFI_REQUIRE(p->mode >= 0 && p->mode <= 2, ATMVFI_EINVAL, "gemm: bad mode %d", p->mode);
    ATMVFI_REQUIRE(p->Cin > 0 && p->Cout > 0, ATMVFI_EINVAL, "gemm: bad channel counts");
    ATMVFI_REQUIRE(planes || (p->in_ld >= atmvfi::round_up(p->Cin, 4) && p->in_ld % 4 == 0), ATMVFI_EALIGN,
                   "gemm: in_ld %d must be a multiple of 4 and >= Cin rounded to 4 (Cin %d)", p->in_ld, p->Cin);
    ATMVFI_REQUIRE(!p->out || (p->out_ld % 4 == 0 && p->out_ld >= atmvfi::round_up(p->Cout, 4)), ATMVFI_EALIGN,
                   "gemm: out_ld %d must be a multiple of 4 and >= Cout rounded to 4 (Cout %d)", p->out_ld, p->Cout);
    ATMVFI_REQUIRE((planes || atmvfi::aligned16(p->in)) && (!p->out || atmvfi::aligned16(p->out)) && atmvfi::aligned16(p->weight), ATMVFI_EALIGN,
                   "gemm: in/out/weight must be 16-byte aligned");
    ATMVFI_REQUIRE((planes || p->in_gstride % 4 == 0) && p->out_gstride % 4 == 0, ATMVFI_EALIGN, "gemm: group strides must be multiples of 4");
    if (p->residual)
        ATMVFI_REQUIRE(p->res_ld >= p->Cout, ATMVFI_EINVAL, "gemm: res_ld %d < Cout %d", p->res_ld, p->Cout);
    if (p->in_prelu) ATMVFI_REQUIRE(atmvfi::aligned16(p->in_prelu), ATMVFI_EALIGN, "gemm: in_prelu must be 16-byte aligned");
    ATMVFI_REQUIRE((!p->bias || atmvfi::aligned16(p->bias)) && (!p->prelu || atmvfi::aligned16(p->prelu)) &&
                       (!p->residual || (atmvfi::aligned16(p->residual) && p->res_ld % 4 == 0)),
                   ATMVFI_EALIGN, "gemm: bias/prelu/residual must be 16-byte aligned (res_ld a multiple of 4)");

    GemmDev d;
    d.mode = p->mode;
    d.in = p->in; d.in_ld = p->in_ld; d.H = p->H; d.W = p->W; d.Cin = p->Cin;
    d.in_gstride = p->in_gstride; d.in_rpg = p->in_rpg;
    d.weight = p->weight;
    d.cin_pad = atmvfi::round_up(p->Cin, 16);
    d.cpt = d.cin_pad / 16;
    d.Cout = p->Cout; d.coutp = atmvfi::round_up(p->Cout, 4);
    d.kw = 1; d.stride = 1; d.pad = 0; d.dil = 1; d.Ho = p->Ho; d.Wo = p->Wo;
    int taps = 1;
    int ngemm = p->Cout;
    if (p->mode == ATMVFI_GEMM_CONV) {
        ATMVFI_REQUIRE(p->N > 0 && p->H > 0 && p->W > 0, ATMVFI_EINVAL, "conv2d: bad input size");
        ATMVFI_REQUIRE((p->kh == 1 || p->kh == 3) && p->kh == p->kw, ATMVFI_EINVAL, "conv2d: kernel %dx%d unsupported", p->kh, p->kw);
        ATMVFI_REQUIRE(p->stride >= 1 && p->dil >= 1 && p->pad >= 0, ATMVFI_EINVAL, "conv2d: bad stride/dilation/pad");
        const int ho = (p->H + 2 * p->pad - p->dil * (p->kh - 1) - 1) / p->stride + 1;
        const int wo = (p->W + 2 * p->pad - p->dil * (p->kw - 1) - 1) / p->stride + 1;
        ATMVFI_REQUIRE(ho == p->Ho && wo == p->Wo, ATMVFI_EINVAL, "conv2d: Ho/Wo (%d,%d) do not match geometry (%d,%d)", p->Ho, p->Wo, ho, wo);
        ATMVFI_REQUIRE(p->in_rpg == 0 && p->out_rpg == 0 && !p->out_row_map, ATMVFI_EINVAL, "conv2d: row groups/maps are LINEAR-only");
        taps = p->kh * p->kw;
        d.kw = p->kw; d.stride = p->stride; d.pad = p->pad; d.dil = p->dil;
        d.M = (long long)p->N * p->Ho * p->Wo;
        ATMVFI_REQUIRE(p->M == 0 || p->M == d.M, ATMVFI_EINVAL, "conv2d: M mismatch");
    } else if (p->mode == ATMVFI_GEMM_LINEAR) {
        ATMVFI_REQUIRE(p->M > 0, ATMVFI_EINVAL, "linear: M must be > 0");
        d.H = 1; d.W = 1;
        d.M = p->M;
    } else {
        ATMVFI_REQUIRE(p->N > 0 && p->H > 0 && p->W > 0, ATMVFI_EINVAL, "deconv: bad input size");
        ATMVFI_REQUIRE(p->Ho == 2 * p->H && p->Wo == 2 * p->W, ATMVFI_EINVAL, "deconv: Ho/Wo must be 2H/2W");
        ATMVFI_REQUIRE(p->in_rpg == 0 && p->out_rpg == 0 && !p->out_row_map && !p->residual, ATMVFI_EINVAL, "deconv: unsupported option");
        d.M = (long long)p->N * p->H * p->W;
        ngemm = 4 * d.coutp;
    }
    d.wrows = packed_rows(p->mode, p->Cout);
    d.ktot = taps * d.cin_pad;
    d.nchunks = taps * d.cpt;
    d.out = p->out; d.out_ld = p->out_ld; d.out_gstride = p->out_gstride; d.out_rpg = p->out_rpg;
    d.out_row_map = p->out_row_map;
    d.bias = p->bias; d.prelu = p->prelu; d.in_prelu = p->in_prelu; d.residual = p->residual; d.res_ld = p->res_ld;
    ATMVFI_REQUIRE(d.M < (1ll << 40), ATMVFI_EINVAL, "gemm: M too large");
    d.a_hi = (const _Float16*)p->in_hi;
    d.a_lo = (const _Float16*)p->in_lo;
    d.a_hi2 = (const _Float16*)p->in_hi2;
    d.a_lo2 = (const _Float16*)p->in_lo2;
    d.in_ld2 = p->in_ld2;
    d.split_chunks = p->in_hi2 ? p->in_split_chunks : (1 << 30);
    d.in_N = p->N;
    d.out_hi = (_Float16*)p->out_hi;
    d.out_lo = (_Float16*)p->out_lo;
    d.out_plane_rows = p->out_plane_rows;
    d.out_c0 = p->out_plane_c0;
    d.out_gc = p->out_plane_gc;
    ATMVFI_REQUIRE(p->tile_wn >= -4 && p->tile_wn <= 8 && (p->tile_wn >= 0 || planes), ATMVFI_EINVAL,
                   "gemm: tile_wn 0 (auto), 1..8, or (split-plane input) -1 reference schedule / -2 gemm_duo (128-column tiles) / -3 gemm_pp / -4 gemm_duo (64-column tiles), got %d", p->tile_wn);
    d.force_wn = p->tile_wn;
    ATMVFI_REQUIRE(!p->workspace || (atmvfi::aligned16(p->workspace) && p->workspace_floats > 0), ATMVFI_EALIGN,
                   "gemm: the split-K workspace must be 16-byte aligned and non-empty");
    d.ksplit = 0;
    d.part = p->workspace;
    d.part_stride = p->workspace ? p->workspace_floats : 0;      // (launch_gemm_split turns the capacity into the real stride)
    d.part_ld = 0;
    d.fit32 = 0;
    d.pfit32 = 0;
    d.nblocks = 0;
    d.dbg = 0;
    d.vblocks = 0;
    d.mchunk = 0;
    d.stamp = nullptr;
    d.w_hi = (const _Float16*)p->weight_hi;
    d.w_lo = (const _Float16*)p->weight_lo;
    d.cin_pad32 = atmvfi::round_up(p->Cin, 32);
    d.cpt32 = d.cin_pad32 / 32;
    d.nchunks32 = taps * d.cpt32;
    d.ktot32 = taps * d.cin_pad32;
    if (p->precision == ATMVFI_PREC_F16X3) {
        ATMVFI_REQUIRE(p->weight_hi && p->weight_lo, ATMVFI_EINVAL, "gemm: precision f16x3 needs weight_hi/weight_lo");
        ATMVFI_REQUIRE(atmvfi::aligned16(p->weight_hi) && atmvfi::aligned16(p->weight_lo), ATMVFI_EALIGN,
                       "gemm: split weights must be 16-byte aligned");
        if (planes) return atmvfi::launch_gemm_split(d, ngemm, (hipStream_t)stream);
        return atmvfi::launch_gemm_f16x3(d, ngemm, (hipStream_t)stream);
    }

    {
        int rc = 0;
        if (atmvfi::try_launch_conv3x3_small(d, p->kh, &rc, (hipStream_t)stream)) return rc;
    }
    // choose the wave tile width WN (block = 128 x 16*WN): MFMA work scales with the padded tile
    // count, operand traffic per MFMA with (1/BM + 1/BN) -- a narrow tile re-reads the activation
    // panel once per n-block (measured: N=197 as 13 x WN=1 ran at 30 TF/s, as 2 x WN=7 at ~90).
    const int ntiles = (ngemm + 15) / 16;
    int best = 1;
    float best_cost = 1e30f;
    for (int wn = 1; wn <= 8; ++wn) {
        const int padded = (ntiles + wn - 1) / wn * wn;
        const float cost = (float)padded * (1.0f + 1.0f / (float)wn);
        if (cost <= best_cost) { best_cost = cost; best = wn; }
    }
    hipStream_t s = (hipStream_t)stream;
    switch (best) {
        case 1: launch<2, 1>(d, ntiles, s); break;
        case 2: launch<2, 2>(d, ntiles, s); break;
        case 3: launch<2, 3>(d, ntiles, s); break;
        case 4: launch<2, 4>(d, ntiles, s); break;
        case 5: launch<2, 5>(d, ntiles, s); break;
        case 6: launch<2, 6>(d, ntiles, s); break;
        case 7: launch<2, 7>(d, ntiles, s); break;
        default: launch<2, 8>(d, ntiles, s); break;
    }
    return atmvfi::check_launch("gemm");
}

extern "C" int atmvfi_conv2d(const atmvfi_gemm_params* p, void* stream) {
    ATMVFI_REQUIRE(p && p->mode == ATMVFI_GEMM_CONV, ATMVFI_EINVAL, "conv2d: params->mode must be ATMVFI_GEMM_CONV");
    return atmvfi_gemm(p, stream);
}
extern "C" int atmvfi_linear(const atmvfi_gemm_params* p, void* stream) {
    ATMVFI_REQUIRE(p && p->mode == ATMVFI_GEMM_LINEAR, ATMVFI_EINVAL, "linear: params->mode must be ATMVFI_GEMM_LINEAR");
    return atmvfi_gemm(p, stream);
}
extern "C" int atmvfi_deconv2x2(const atmvfi_gemm_params* p, void* stream) {
    ATMVFI_REQUIRE(p && p->mode == ATMVFI_GEMM_DECONV, ATMVFI_EINVAL, "deconv2x2: params->mode must be ATMVFI_GEMM_DECONV");
    return atmvfi_gemm(p, stream);
}

/*
 * atmvfi.h -- C ABI of libatmvfi_hip.so: the MI355X (gfx950) kernels underneath the
 * ATM-VFI `Network.forward` hot path.
 *
 * The reference (Gancheekim/ATM-VFI) has no FFI seam: every device op is an implicit
 * ATen call issued from Python.  Each entry point below therefore
 * cites the reference *call site(s)* whose arithmetic it replaces.  The Python host
 * (atm-vfi_amd/hip_ops.py) binds these with ctypes; INTEGRATION.md shows the binding.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to fp32 unless stated; the library allocates
 *    nothing, owns nothing and keeps no mutable global state (re-entrant per stream; the only caches are per-device
 *    constants: CU count and "dynamic LDS limit already raised for this kernel"); it reads no environment variable -- the
 *    diagnostic builds (`make stamp`, `make ablate`) do, and are never loaded by the product;
 *  - `stream` is a hipStream_t passed as void* (0 = default stream);
 *  - feature maps are NHWC "views": (ptr, ld) with ptr already advanced to the first
 *    channel of the view and `ld` = floats between consecutive pixels.  Views let a
 *    producer write straight into a slice of a wider concat buffer, so torch.cat /
 *    einops.rearrange of the reference never materialise.  ld % 4 == 0 and 16-byte
 *    aligned pointers are required (checked; ATMVFI_EALIGN);
 *  - image-like tensors (3-channel frames, 2-channel flows, 1-channel masks) are
 *    planar NCHW, exactly as the reference's API returns them;
 *  - return value: 0 on success, negative ATMVFI_E* otherwise; atmvfi_last_error()
 *    gives a thread-local message.  Shapes are validated on the host BEFORE launch.
 */
#ifndef ATMVFI_H
#define ATMVFI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ATMVFI_OK        0
#define ATMVFI_EINVAL   -1   /* bad shape / null pointer / unsupported configuration */
#define ATMVFI_EALIGN   -2   /* pointer or leading dimension not 16-byte aligned */
#define ATMVFI_ELAUNCH  -3   /* HIP launch error */

int         atmvfi_version(void);            /* (major<<16)|(minor<<8)|patch */
const char* atmvfi_last_error(void);
/* sha256 (hex) over the sources this library was built from (tools/source_digest.py: atm-vfi_amd/csrc/{*.hip,*.h,*.inc,Makefile} +
 * this header), baked in at build time: lets a caller prove that a shipped .so belongs to the shipped sources.  The reference has
 * no counterpart (pure Python, nothing is built). */
const char* atmvfi_source_digest(void);
/* The f16x3 engines' operand range contract at run time.  Contraction operands are split as x = hi + lo'/1024 in fp16: |x| beyond
 * 65504 saturates silently (finite, but no longer what the fp32 reference computes).  The CHECKED build of this library
 * (libatmvfi_hip_checked.so: the same sources with -DATMVFI_RANGE_CHECK, `make checked`) counts, in a device word the caller
 * attaches, every pair of activations whose split produced a hi half at the fp16 limit or non-finite (|x| >= 65488, inf, NaN) --
 * at every place an activation is split: the plane sinks of all producers and the in-kernel operand splits.  `word`: a zeroed
 * device uint32 that must outlive the launches (NULL detaches); attached for the CURRENT device, ordered on `stream`.
 * atmvfi_range_checked(): 1 in the checked build, 0 in the default one, where atmvfi_range_word_set fails with ATMVFI_EINVAL
 * and no kernel carries an extra instruction.  The reference has no counterpart (it computes in fp32). */
int atmvfi_range_word_set(uint32_t* word, void* stream);
int atmvfi_range_checked(void);

/* ------------------------------------------------------------------------------------
 * Implicit-GEMM contraction engine on fp32 MFMA (v_mfma_f32_16x16x4_f32).
 *
 * One kernel family serves the three contraction shapes of the hot path:
 *   ATMVFI_GEMM_CONV   Conv2d k in {1,3}, stride {1,2,4}, dilation {1,2}, zero pad
 *                      (network_base.py:20-25 conv(); :42-48 fusion convs; :158,:195 heads)
 *   ATMVFI_GEMM_LINEAR nn.Linear over token rows (attention.py:138-141,349-351 q/kv/qkv/proj;
 *                      :93-96 fc1/fc2), with optional row groups / row scatter map
 *   ATMVFI_GEMM_DECONV ConvTranspose2d(k2,s2,p0) (network_base.py:27-32 deconv()) as a GEMM to
 *                      4*Cout columns with a pixel-shuffle store
 * Epilogue, in order: + bias[co]; PReLU(slope[co]) if `prelu`; + residual if `residual`.
 * `in_prelu` applies a per-input-channel PReLU while loading the input (the leading
 * nn.PReLU of upsample_pyramid stages 1-2, network_base.py:209,215).
 * ---------------------------------------------------------------------------------- */
#define ATMVFI_GEMM_CONV   0
#define ATMVFI_GEMM_LINEAR 1
#define ATMVFI_GEMM_DECONV 2

typedef struct atmvfi_gemm_params {
    int32_t mode;
    /* input view.  CONV/DECONV: NHWC [N,H,W,Cin].  LINEAR: rows; row m lives at
       in + (m / in_rpg) * in_gstride + (m % in_rpg) * in_ld   (in_rpg == 0: m * in_ld) */
    const float* in;
    int32_t in_ld, N, H, W, Cin;
    int64_t in_gstride; int32_t in_rpg;
    /* weights packed by atmvfi_pack_*: [Nrows16][taps][CinPad16] fp32, zero padded */
    const float* weight;
    int32_t Cout, kh, kw, stride, pad, dil;
    int32_t Ho, Wo;            /* CONV: output size; DECONV: 2H,2W; LINEAR: ignored */
    int64_t M;                 /* LINEAR: number of rows; others: N*Ho*Wo (CONV) / N*H*W (DECONV) */
    /* output view; row r at out + (r / out_rpg) * out_gstride + (r % out_rpg) * out_ld */
    float* out;
    int32_t out_ld;
    int64_t out_gstride; int32_t out_rpg;
    const int32_t* out_row_map;   /* optional [M]: destination row per GEMM row, <0 = drop */
    const float* bias;            /* [Cout] or NULL */
    const float* prelu;           /* [Cout] or NULL */
    const float* in_prelu;        /* [Cin]  or NULL */
    const float* residual;        /* optional, indexed by GEMM row m (before out_row_map) */
    int32_t res_ld;
    /* ATMVFI_PREC_F32: exact fp32 MFMA on `weight`.  ATMVFI_PREC_F16X3: split-precision engine
       (x = hi + lo'/1024 in fp16, three 16-bit MFMAs per product, fp32 accumulate) on the two
       fp16 planes written by atmvfi_pack_weight_split; in_prelu must then be padded to 32. */
    int32_t precision;
    const void* weight_hi;
    const void* weight_lo;
    /* Split-plane input (F16X3; every mode, CONV: see in_hi2 below): when in_hi/in_lo are set the activations are read by LDS-DMA from two
       fp16 planes in the chunk-major layout [Cin/32 chunks][plane rows][32] written by a producer kernel's sink or by
       atmvfi_split_planes; `in_ld` is then the plane row count (>= M), `in` is ignored, the pad channels of the last chunk must be
       finite (they meet zero weights). */
    const void* in_hi;
    const void* in_lo;
    /* Plane sink (F16X3 only): when out_hi/out_lo are set the result is written (also, or -- with out == NULL -- only) as split
       planes in the same chunk-major layout with out_plane_rows rows per chunk: channel co of output row r goes to plane row
       r % out_rpg, channel out_plane_c0 + (r / out_rpg) * out_plane_gc + co (out_rpg == 0: row r, channel out_plane_c0 + co;
       DECONV: r = output pixel n*Ho*Wo + y*Wo + x; with out_row_map, r is the mapped row).  Channel offsets are multiples of 4;
       channels past Cout inside the last group of 4 are written as zero.  This is the input format of atmvfi_conv3x3_planes
       and of atmvfi_gemm's in_hi/in_lo: consecutive contraction layers hand activations over without an fp32 copy. */
    void* out_hi;
    void* out_lo;
    int64_t out_plane_rows;
    int32_t out_plane_c0, out_plane_gc;
    /* F16X3, fp32 input: n-tiles of 16 columns per 256-row workgroup tile, 0 = cost model (default), 1..8 = forced (sweeps).
       F16X3, split-plane input: 0 = chosen per launch between the ping-pong kernel (gemm_pp.hip: 256 x 128 tiles, persistent grid) and
       gemm_duo.hip (128 x 128 tiles, two workgroups per CU: grids that would leave half of the CUs idle; 128 x 64 tiles for layers of
       at most 64 columns); -3 forces gemm_pp.hip, -2 / -4 gemm_duo.hip with 128- / 64-column tiles, -1 the reference schedule
       (gemm_split.hip).  All have the same arithmetic: bit-identical results (parity tests, same-process A/B). */
    int32_t tile_wn;
    /* CONV mode on split-plane input (in_hi / in_lo): rows are input pixels n*H*W + y*W + x, in_ld > N*H*W and row N*H*W of every chunk
       is zero (taps outside the image read it).  Optionally the input channels come from TWO plane buffers (a torch.cat of the
       reference that never materialises): chunks 0 .. in_split_chunks-1 from in_hi / in_lo, the rest from in_hi2 / in_lo2 (in_ld2 rows
       per chunk, same pixel rows, same zero-row contract); both parts are whole 32-channel chunks. */
    const void* in_hi2;
    const void* in_lo2;
    int32_t in_ld2, in_split_chunks;
    /* Split-K scratch (F16X3, split-plane input; optional): with `workspace` (fp32, 16-byte aligned) of at least
       atmvfi_gemm_workspace_floats(M, ngemm, k-steps) floats the launcher may cut a long K of an under-filled grid (small frames:
       a few tiles walking 40-150 k-steps on a mostly idle chip) into up to 8 ranges of k-steps, one workgroup per (tile, range)
       writing raw fp32 partial sums, and a second kernel that adds them IN RANGE ORDER and runs the epilogue.  Run-to-run
       deterministic; differs from the unsplit launch by fp32 summation order only.  NULL: never split. */
    float* workspace;
    int64_t workspace_floats;
} atmvfi_gemm_params;

#define ATMVFI_PREC_F32   0
#define ATMVFI_PREC_F16X3 1

int atmvfi_gemm(const atmvfi_gemm_params* p, void* stream);
/* fp32 elements of split-K scratch the plane-input GEMM wants for M rows, ngemm GEMM columns (Cout; 4 * round_up(Cout, 4) for DECONV)
 * and ksteps = taps * ceil(Cin / 32) k-steps on this device; 0: it would not split that shape. */
int64_t atmvfi_gemm_workspace_floats(int64_t M, int ngemm, int ksteps);

/* 1x1 convolution with 1 <= Cout <= 8 output channels on split-plane input: the read-out nn.Conv2d(hidden, 5, 1) of the motion MLPs
 * (network_base.py:158,195 local_motion_mlp[2] / global_motion_mlp[2]).  in_hi / in_lo: chunk-major planes of `rows` pixel rows
 * (plane_rows rows per 32-channel chunk); weight: the layer's fp32 [Cout][Cin] (the nn.Conv2d weight as it is); out: fp32 rows,
 * out_ld floats apart, Cout written per row.  fp32 FMAs on x = hi + lo'/1024. */
int atmvfi_head1x1_planes(const void* in_hi, const void* in_lo, int64_t plane_rows, int64_t rows, int Cin, const float* weight,
                          const float* bias, int Cout, float* out, int out_ld, void* stream);

/* fp32 rows [M, C] (row stride in_ld floats) -> the two fp16 planes of the split-plane format, hi = fp16(x),
   lo = fp16((x - hi) * 1024), both saturating, chunk major: element (row, c) at ((c / 32) * plane_rows + row) * 32 + c % 32
   (each plane holds ceil(C / 32) * plane_rows * 32 halves; the pad channels of the last chunk are written as zero).
   `prelu` (optional, [C]) is applied to the values first. */
int atmvfi_split_planes(const float* in, int in_ld, int64_t M, int C, const float* prelu, void* hi, void* lo, int plane_rows, void* stream);
/* The same into a channel range of wider planes: the C channels go to plane channels c0 .. c0 + C (c0 a multiple of 8), channels up to
   the next multiple of pad_to (8, or 32 with c0 a multiple of 32) are written as zero, everything else is left alone. */
int atmvfi_split_planes_at(const float* in, int in_ld, int64_t M, int C, const float* prelu, void* hi, void* lo, int plane_rows, int c0,
                           int pad_to, void* stream);

/* Convenience wrappers with the reference-layer names (thin shims over atmvfi_gemm). */
int atmvfi_conv2d(const atmvfi_gemm_params* p, void* stream);
int atmvfi_linear(const atmvfi_gemm_params* p, void* stream);
int atmvfi_deconv2x2(const atmvfi_gemm_params* p, void* stream);

/* Weight re-layout (device -> device, run once per checkpoint load).
 *  conv  : OIHW [Cout,Cin,kh,kw]  -> [rows16(Cout)][kh*kw][CinPad16]
 *  linear: [Cout,Cin]             -> same with one tap
 *  deconv: IOHW [Cin,Cout,2,2]    -> [rows16(4*CoutP4)][1][CinPad16], row = (a*2+b)*CoutP4 + co
 * `dst` must hold atmvfi_packed_weight_floats(...) floats. */
int64_t atmvfi_packed_weight_floats(int mode, int Cout, int Cin, int kh, int kw);
int atmvfi_pack_weight(int mode, const float* src, float* dst, int Cout, int Cin, int kh, int kw, void* stream);

/* ------------------------------------------------------------------------------------
 * Split-precision ("f16x3") 3x3 / stride 1 / pad 1 convolution -- the same Conv2d(+PReLU) call
 * sites as ATMVFI_GEMM_CONV (network_base.py:20-25), two thirds of the network's FLOPs.
 * Every fp32 operand is split as x = hi + lo (fp16 each) and hi*hi + hi*lo + lo*hi is accumulated
 * in fp32 on v_mfma_f32_16x16x32_f16 (3 MFMAs at 16x the fp32-MFMA rate; ~22 significand bits,
 * finite for |x| < 1.3e5).  The input halo of a 16x16 output tile is staged in LDS once per
 * 32-channel chunk and reused by all nine taps.  Epilogue: + bias[co]; PReLU(slope[co]).
 * Split weights of the two GEMM engines (atmvfi_gemm with precision F16X3): atmvfi_pack_weight_split writes two fp16 planes
 * (atmvfi_split_weight_halves() halves each), K-STEP MAJOR: [k-step = tap * CinPad32/32 + chunk][row16][32 halves], same source
 * layouts and row order as atmvfi_pack_weight -- the 16 rows x 64 bytes one load / LDS-DMA instruction moves are one contiguous KiB.
 * ---------------------------------------------------------------------------------- */
int64_t atmvfi_split_weight_halves(int mode, int Cout, int Cin, int kh, int kw);
int atmvfi_pack_weight_split(int mode, const float* src, void* dst_hi, void* dst_lo, int Cout, int Cin, int kh, int kw,
                             void* stream);
/* Weights of atmvfi_conv3x3_f16x3 (OIHW [Cout,Cin,3,3] -> two fp16 planes of atmvfi_conv3x3_weight_halves halves each), k-step
 * major: [k-step][row padded to 16][32 halves], k-step = (32-channel chunk, tap); when 1 <= Cin % 32 <= 8 the channel tail is
 * tap-packed into three more k-steps of 4 taps x 8 channels, so that e.g. the 32k+5-wide decoder maps do not pay nine 32-wide
 * k-steps for five channels.  Layout details: csrc/conv3x3_f16x3.hip. */
int64_t atmvfi_conv3x3_weight_halves(int Cout, int Cin);
int atmvfi_pack_weight_conv3x3(const float* src, void* dst_hi, void* dst_lo, int Cout, int Cin, void* stream);
/* Optional second output (out_hi / out_lo non-null): the result once more as split planes -- fp16 hi and lo' = (x - hi) * 1024,
 * chunk major [ceil(Cout/32)][plane_rows][32] with pixel n*H*W + y*W + x as the row, see atmvfi_split_planes -- after a
 * per-channel PReLU of its own (plane_prelu, padded to a multiple of 32 floats, or null).  That is the operand format of the
 * LDS-DMA GEMM behind atmvfi_deconv2x2 / atmvfi_linear (in_hi / in_lo): the decoder's conv -> PReLU -> ConvTranspose2d chain
 * (network_base.py:207-216) hands over without a separate split pass.  Pad channels of the last chunk must be zero in the
 * caller's buffer; those inside the last group of 4 are written as zero. */
int atmvfi_conv3x3_f16x3(const float* in, int in_ld, int N, int H, int W, int Cin, const void* w_hi, const void* w_lo,
                         int Cout, float* out, int out_ld, const float* bias, const float* prelu, void* out_hi, void* out_lo,
                         int64_t plane_rows, const float* plane_prelu, int schedule, int wn, void* stream);
/* schedule / wn: the kernel has two schedules (512-thread 16x16 tiles, one workgroup per CU; 256-thread 16x8 tiles, two per CU) and
 * 1..8 n-tiles of 16 output channels per workgroup, picked per layer from a cost model.  Per call: schedule -1 = cost model,
 * 0 = row, 1 = half; wn 0 = cost model, 1..8 = n-tiles per workgroup (parity tests reach every instance this way; results do not
 * depend on either).  The library keeps no process-wide override. */
/* The same convolution on SPLIT-PLANE input (csrc/conv3x3_planes.hip): the activations are the two fp16 planes (hi, lo', chunk major
 * [ceil(Cin/32)][in_rows][32], row = pixel n*H*W + y*W + x) that the producing layer's sink wrote, so the input halo goes
 * global -> LDS by LDS-DMA with no register staging and no conversion, and the two waves of each SIMD run one phase apart (one
 * issues MFMAs while the other reads fragments and issues DMA).  Bit-identical to atmvfi_conv3x3_f16x3 on the same values.
 * Contract of the input planes: in_rows > N*H*W and row N*H*W of every chunk is ZERO (it is the source of the halo pixels that
 * fall outside the image; atm-vfi_amd's Planes.alloc reserves it); pad channels of the last chunk are zero; in_rows * 64 < 2^32.
 * Outputs: fp32 NHWC view `out` (may be NULL; only channels >= out_cmin, a multiple of 4, are stored: columns below are
 * never touched, so `out` may point out_cmin floats in front of a compact buffer of just the stored channels) and / or the plane sink
 * out_hi / out_lo (may be NULL; not both NULL) at channel offset out_c0 (multiple of 8) of a plane buffer with plane_rows rows,
 * through plane_prelu (as atmvfi_conv3x3_f16x3); plane channels from Cout up to the next multiple of 8 are written as zero.
 * wn: 0 = pick the tile width (n-tiles of 16 output channels per workgroup) from the cost model, 1..8 = force it (tests, sweeps). */
int atmvfi_conv3x3_planes(const void* in_hi, const void* in_lo, int64_t in_rows, int N, int H, int W, int Cin, const void* w_hi,
                          const void* w_lo, int Cout, float* out, int out_ld, const float* bias, const float* prelu, void* out_hi,
                          void* out_lo, int64_t plane_rows, int out_c0, const float* plane_prelu, int out_cmin, int wn, void* stream);
/* The same with a SECOND, raw plane sink (out_hi2 / out_lo2, plane_rows2 rows per chunk, channel offset out_c02, a multiple of 8; needs
 * the first sink): a decoder map of the reference goes on twice -- through the next stage's leading nn.PReLU into that stage's deconv
 * (network_base.py:209,215: the first sink with plane_prelu) and as it is into the refiner's strided convs (network_base.py:421-424,
 * torch.cat([feat, dec], 1): atmvfi_gemm CONV mode with in_hi2 / in_lo2 reads it). */
int atmvfi_conv3x3_planes2(const void* in_hi, const void* in_lo, int64_t in_rows, int N, int H, int W, int Cin, const void* w_hi,
                           const void* w_lo, int Cout, float* out, int out_ld, const float* bias, const float* prelu, void* out_hi,
                           void* out_lo, int64_t plane_rows, int out_c0, const float* plane_prelu, void* out_hi2, void* out_lo2,
                           int64_t plane_rows2, int out_c02, int out_cmin, int wn, void* stream);
/* The same with SPLIT-K for launches whose tiles leave most of the chip idle (small frames: the motion-MLP layers of network_lite at
 * 256 x 256 are 22 workgroups walking 200 k-steps each): with a `workspace` of at least atmvfi_conv3x3_planes_workspace_floats() floats
 * (16-byte aligned; NULL / too small: no split) the K range is cut into up to 8 ranges of whole 32-channel chunks, one workgroup per
 * (tile, range) writes raw fp32 partial sums into the workspace and a second kernel adds them IN RANGE ORDER and runs the epilogue
 * (bias, PReLU, fp32 rows, plane sinks).  Run-to-run deterministic; differs from the unsplit launch by fp32 summation order only
 * (the reference's conv, network_base.py:20-25, fixes no order either).  atmvfi_conv3x3_planes_workspace_floats returns 0 when the
 * launcher would not split that shape on this device. */
int64_t atmvfi_conv3x3_planes_workspace_floats(int N, int H, int W, int Cin, int Cout);
int atmvfi_conv3x3_planes3(const void* in_hi, const void* in_lo, int64_t in_rows, int N, int H, int W, int Cin, const void* w_hi,
                           const void* w_lo, int Cout, float* out, int out_ld, const float* bias, const float* prelu, void* out_hi,
                           void* out_lo, int64_t plane_rows, int out_c0, const float* plane_prelu, void* out_hi2, void* out_lo2,
                           int64_t plane_rows2, int out_c02, int out_cmin, int wn, float* workspace, int64_t workspace_floats, void* stream);
/* ------------------------------------------------------------------------------------
 * LayerNorm over the channel axis of token rows (eps 1e-5, affine), optional gather.
 * Replaces nn.LayerNorm at attention.py:316 (norm1 on windowed tokens), :333 (norm2) and
 * network_base.py:84 (fusion norm).  With `src_row_map` (length `rows`), output row r reads
 * source row src_row_map[r]; a negative entry denotes a zero-padded token, whose
 * LayerNorm is exactly `beta` (attention.py:58-61 pads with zeros BEFORE norm1).
 * The map fuses pad_if_needed + torch.roll + window_partition (attention.py:273-313).
 * Output sinks (also of atmvfi_dwconv3x3_gelu and atmvfi_window_attention): fp32 rows `out`, and/or the split-plane
 * pair `out_hi`/`out_lo` (fp16, chunk major with `plane_rows` rows per 32-channel chunk) that feeds atmvfi_gemm's in_hi/in_lo
 * directly -- the consumer GEMM then skips its fp32 -> fp16-pair conversion.  Either side may be NULL, not both.
 * ---------------------------------------------------------------------------------- */
int atmvfi_layernorm(const float* in, int in_ld, int64_t in_gstride, int in_rpg,
                     const int32_t* src_row_map, float* out, int out_ld,
                     const float* gamma, const float* beta, int64_t rows, int C,
                     void* out_hi, void* out_lo, int plane_rows, void* stream);

/* Depth-wise 3x3 conv (pad 1, bias) + exact GELU on NHWC tokens: DWConv + act of
 * attention.py:74-85,118-119.  weight9 is [9][C] (tap-major), see atmvfi_pack_dw_weight. */
int atmvfi_dwconv3x3_gelu(const float* in, int in_ld, float* out, int out_ld,
                          const float* weight9, const float* bias,
                          int N, int H, int W, int C, void* out_hi, void* out_lo, int plane_rows, void* stream);
int atmvfi_pack_dw_weight(const float* src /*[C,1,3,3]*/, float* dst /*[9][C]*/, int C, void* stream);

/* ------------------------------------------------------------------------------------
 * Fused window attention (attention.py:187-213 AttentionToMotion.forward, :370-390
 * WindowAttention.forward).  qkv: [Bw*N, 3C] rows in window order, columns [q | k | v],
 * head h at columns h*hd.  One workgroup per (window, head): K/V tiles staged in LDS,
 * S^T = K Q^T and O^T = V^T P^T on fp32 MFMA, softmax by wavefront reductions; the
 * N x N attention matrix never reaches HBM.
 *   labels : optional [nW, N] int32 per-token region label; additive mask is
 *            -100 * (label_q != label_k)  (attention.py:54-57, 296-303); window index of
 *            block b is b % nW.
 *   kv_shift: K/V are read from window (b + kv_shift) % Bw -- Bw/2 for the cross-frame
 *            attention of ATMFormer (attention.py:318), 0 for self attention.
 *   motion : optional [Bw, N, heads, 2]: per-head expected key offset
 *            sum_k A[q,k] * (k_xy - q_xy)   (attention.py:207-208)
 * ---------------------------------------------------------------------------------- */
int atmvfi_window_attention(const float* qkv, float* out /*[Bw*N, C]*/, float* motion,
                            const int32_t* labels, int Bw, int nW, int ws, int heads, int hd,
                            int kv_shift, void* out_hi, void* out_lo, int plane_rows, void* stream);
/* The same contract on the f16x3 arithmetic of the contraction engines (x = hi + lo'/1024 in fp16, three
 * v_mfma_f32_16x16x32_f16 per product, fp32 accumulation: ~22 significand bits; operands beyond +-65504 saturate like
 * every other f16x3 kernel): K / V converted while they are staged, V read back through gfx950's transposing LDS read.
 * What Network.forward calls under precision "f16x3"; atmvfi_window_attention stays the exact-fp32 kernel. */
int atmvfi_window_attention_f16x3(const float* qkv, float* out /*[Bw*N, C]*/, float* motion,
                                  const int32_t* labels, int Bw, int nW, int ws, int heads, int hd,
                                  int kv_shift, void* out_hi, void* out_lo, int plane_rows, void* stream);
/* Named entry points of SURVEY.md section 8b (shims over atmvfi_window_attention). */
int atmvfi_window_attn_cross_motion(const float* qkv, float* out, float* motion, const int32_t* labels,
                                    int Bw, int nW, int ws, int heads, int hd, void* stream);
int atmvfi_window_attn_self(const float* qkv, float* out, const int32_t* labels,
                            int Bw, int nW, int ws, int heads, int hd, void* stream);

/* Head read-out of the motion (attention.py:143-146, 209-211): Linear(heads->heads/2), GELU,
 * Linear(heads/2->1) over the head axis, separately for dx and dy, then scatter from window
 * order to image order (window_reverse + roll back + de-pad, attention.py:324-331) and into
 * the channel-stacked motion map of the motion MLP input (network_base.py:377-382):
 * destination = out + (r / out_rpg) * out_gstride + (r % out_rpg) * out_ld + {0,1}, r = row_map[m]. */
int atmvfi_motion_head(const float* motion /*[rows, heads, 2]*/, const int32_t* row_map,
                       const float* w0 /*[heads/2, heads]*/, const float* b0, const float* w1 /*[1, heads/2]*/,
                       const float* b1, float* out, int out_ld, int64_t out_gstride, int out_rpg,
                       int64_t rows, int heads, void* stream);
/* The same, and the two values again as split planes (hi / lo', chunk major, plane_rows rows per chunk): row r = row_map[m] goes to plane
 * row r % out_rpg, channels plane_c0 + (r / out_rpg) * plane_gc + {0, 1} -- the eight motion channels of the motion MLP's plane input
 * (network_base.py:377-382, 410) without a separate split pass.  Channel offsets must be even. */
int atmvfi_motion_head_planes(const float* motion, const int32_t* row_map, const float* w0, const float* b0, const float* w1,
                              const float* b1, float* out, int out_ld, int64_t out_gstride, int out_rpg, int64_t rows, int heads,
                              void* out_hi, void* out_lo, int64_t plane_rows, int plane_c0, int plane_gc, void* stream);

/* ------------------------------------------------------------------------------------
 * Backward bilinear warp, zero padding (flow_warp.py:50-60 -> grid_sample(bilinear, zeros,
 * align_corners=True)); coordinates are generated in-kernel (the reference builds the
 * grid on the CPU on every call, flow_warp.py:57).
 * Flow is addressed as fx = flow[b*flow_bstride + pix*flow_pstride], fy = fx[flow_cstride],
 * which covers planar [B,2,H,W] (pstride 1, cstride H*W) and the last-5-channel NHWC
 * motion maps (pstride ld, cstride 1).
 * ---------------------------------------------------------------------------------- */
int atmvfi_flow_warp(const float* src /*[B,C,H,W]*/, const float* flow, int64_t flow_bstride,
                     int flow_pstride, int flow_cstride, float* dst, int B, int C, int H, int W, void* stream);
/* atmvfi_flow_warp with LDS-STAGED SOURCE TILES (the same signature and bit-identical results): a workgroup owns 32 x 8 output pixels,
 * stages the bounding box of their bilinear taps (up to 64 x 24 source pixels per plane, 16-byte row loads) in LDS and takes the taps
 * from there; tiles whose flows reach further gather from global memory.  Needs W % 4 == 0 and a 16-byte aligned src (ATMVFI_EINVAL
 * otherwise: the caller uses atmvfi_flow_warp). */
int atmvfi_flow_warp_tiled(const float* src /*[B,C,H,W]*/, const float* flow, int64_t flow_bstride,
                           int flow_pstride, int flow_cstride, float* dst, int B, int C, int H, int W, void* stream);
/* The reference's non-default forms of flow_warp (flow_warp.py:50-60 with mask=True and / or padding_mode != 'zeros';
 * bilinear_sample :26-47): planar flow [B,2,H,W]; padding_mode 0 'zeros', 1 'border', 2 'reflection' (grid_sample's coordinate maps,
 * align_corners=True); mask (may be NULL): [B,H,W] bytes, 1 where the normalised sampling coordinate lies in [-1, 1] on both axes
 * (flow_warp.py:43, on the reference's own fp32 expression).  Not on the forward's path. */
int atmvfi_flow_warp_ex(const float* src /*[B,C,H,W]*/, const float* flow /*[B,2,H,W]*/, float* dst, uint8_t* mask /*[B,H,W] or NULL*/,
                        int B, int C, int H, int W, int padding_mode, void* stream);
int atmvfi_flow_warp_nhwc(const float* src, int src_ld, int64_t src_bstride,
                          const float* flow, int64_t flow_bstride, int flow_pstride, int flow_cstride,
                          float* dst, int dst_ld, int64_t dst_bstride, int B, int C, int H, int W, void* stream);

/* Fused per-level synthesis (network_base.py:464-466, 496-498, 523-525): warp both frames with
 * their flows, mask = sigmoid(logit), I_t = mask*I0w + (1-mask)*I1w.  `motion` is a 5-channel
 * map [flow0.xy, flow1.xy, mask logit] with pixel stride `motion_ld` (NHWC slice).  Optional
 * outputs (NULL to skip): planar flows/masks for the returned dict, and a 15-channel NHWC
 * pack [im0, I0w, im1, I1w, I_t] into the refiner input (network_base.py:418; im0/im1 there
 * are the ORIGINAL frames `orig0/orig1`). */
int atmvfi_warp_blend(const float* im0, const float* im1 /*[B,3,H,W] (pre-warped pyramids)*/,
                      const float* motion, int motion_ld, int64_t motion_bstride,
                      float* i0w, float* i1w, float* it /*[B,3,H,W]*/,
                      float* flow0_out, float* flow1_out /*[B,2,H,W]*/, float* mask1_out, float* mask2_out /*[B,1,H,W]*/,
                      const float* orig0, const float* orig1, float* pack15, int pack_ld,
                      int B, int H, int W, void* stream);
/* The same with a plane sink for the pack: the 15 values and one zero as split planes (atmvfi_split_planes layout, pack_rows rows per
 * chunk, row = pixel b*H*W + y*W + x) at channels pack_c0 .. pack_c0 + 16 (pack_c0 a multiple of 4) -- the operand format of
 * atmvfi_conv3x3_planes, which the refiner's first conv runs on.  pack_hi / pack_lo may be NULL (= atmvfi_warp_blend). */
int atmvfi_warp_blend_planes(const float* im0, const float* im1, const float* motion, int motion_ld, int64_t motion_bstride,
                             float* i0w, float* i1w, float* it, float* flow0_out, float* flow1_out, float* mask1_out, float* mask2_out,
                             const float* orig0, const float* orig1, float* pack15, int pack_ld, void* pack_hi, void* pack_lo,
                             int64_t pack_rows, int pack_c0, int B, int H, int W, void* stream);
/* atmvfi_warp_blend_planes with LDS-staged source tiles of both images (see atmvfi_flow_warp_tiled): the same signature, bit-identical
 * results; W % 4 == 0, im0 / im1 16-byte aligned. */
int atmvfi_warp_blend_tiled(const float* im0, const float* im1, const float* motion, int motion_ld, int64_t motion_bstride,
                            float* i0w, float* i1w, float* it, float* flow0_out, float* flow1_out, float* mask1_out, float* mask2_out,
                            const float* orig0, const float* orig1, float* pack15, int pack_ld, void* pack_hi, void* pack_lo,
                            int64_t pack_rows, int pack_c0, int B, int H, int W, void* stream);
#define atmvfi_blend atmvfi_warp_blend   /* SURVEY.md section 8b name */

/* Bilinear resize with align_corners=True, src = dst*(in-1)/(out-1), values * value_scale:
 * F.interpolate(scale 0.5) of the image pyramid (network_base.py:445-446,461-462) and
 * upsample_flow (network_base.py:11-18, value_scale = factor).  Source element (b,c,y,x) is
 * src[b*sb + c*sc + y*sy + x*sx] (planar NCHW or a channel slice of an NHWC motion map);
 * dst is planar contiguous [B,C,Ho,Wo]. */
int atmvfi_resize_bilinear_ac(const float* src, int64_t src_bstride, int64_t src_cstride, int64_t src_ystride,
                              int64_t src_xstride, float* dst, int B, int C, int Hi, int Wi, int Ho, int Wo,
                              float value_scale, void* stream);

/* The three x0.5 levels of the image pyramid of both frames (network_base.py:444-448) in one launch: im0 / im1 planar [B,3,H,W]
 * (H, W multiples of 8); l1 / l2 / l3 planar [2B,3,H>>l,W>>l], frame 0's images first.  Bit-identical to atmvfi_resize_bilinear_ac
 * applied level by level. */
int atmvfi_image_pyramid(const float* im0, const float* im1, float* l1, float* l2, float* l3, int B, int H, int W, void* stream);
/* The same and, with `pack` non-NULL, atmvfi_pack_frames' output [2B,H,W,4] in the same launch (both read only the two frames). */
int atmvfi_image_pyramid_pack(const float* im0, const float* im1, float* l1, float* l2, float* l3, float* pack, int B, int H, int W,
                              void* stream);

/* flow_warp (flow_warp.py:50-60) of a planar image [B,C,H,W] by a contiguous planar flow [B,2,H,W] into dst, AND that flow up-sampled
 * to [B,2,2H,2W] with its values doubled (upsample_flow, network_base.py:11-18) into flow_up, in one launch: one step of the global
 * flow's walk down the image pyramid (network_base.py:468-485).  Bit-identical to atmvfi_flow_warp + atmvfi_resize_bilinear_ac(x2). */
int atmvfi_flow_warp_up2(const float* src, const float* flow, float* dst, float* flow_up, int B, int C, int H, int W, void* stream);
/* The same with the warp half on LDS-staged source tiles (atmvfi_flow_warp_tiled): bit-identical; W % 4 == 0, src 16-byte aligned. */
int atmvfi_flow_warp_up2_tiled(const float* src, const float* flow, float* dst, float* flow_up, int B, int C, int H, int W, void* stream);

/* torch.cat([im0, im1], 0) (network_base.py:451) fused with NCHW -> NHWC4 (4th channel 0). */
int atmvfi_pack_frames(const float* im0, const float* im1, float* dst /*[2B,H,W,4]*/, int B, int H, int W, void* stream);

/* I_t += 2*sigmoid(r) - 1; clamp(0,1) (network_base.py:429,532-533).  `r` is the NHWC 3-channel
 * output of refine_head; writes the unclamped sum (the tensor the reference leaves in
 * im_t_list[0]) and the clamped frame. */
int atmvfi_final_residual(const float* it /*[B,3,H,W]*/, const float* r, int r_ld,
                          float* it_sum, float* it_clamped, int B, int H, int W, void* stream);

/* The tail of the refiner in two launches instead of three, without the 64-channel full-resolution map r1 ever reaching HBM
 * (network_base.py:257-260 refine_head = conv(2c, c) + PReLU, conv(c, 3) + PReLU; :429 2 * sigmoid - 1; :532-533 += , clamp):
 *  - atmvfi_conv3x3_planes_readout: refine_head.0 as atmvfi_conv3x3_planes (Cout = 32 or 64: one column block) whose epilogue multiplies
 *    the activated tile -- still in registers, in the accumulator layout, which is the B-operand layout of the next MFMA -- by the
 *    27 x Cout matrix W2[(tap, o)][c] = refine_head.1's weight [o][c][tap] (f16x3 like every contraction) and stores the 27 per-pixel
 *    "tap contributions" as planar fp32 contrib[(tap * 3 + o) * contrib_plane + pixel].  w2: fp16 [plane hi / lo'][row tile 2]
 *    [k-step Cout / 32][lane 64][8] in the kernel's register order (hip_ops.HipOps.pack_readout builds it);
 *  - atmvfi_refine_tail: out[o][p] = bias[o] + sum over the nine taps of contrib[tap * 3 + o][p + offset(tap)] (taps outside the image
 *    contribute nothing: the convolution's zero padding), PReLU(slope), 2 * sigmoid - 1, + it, and the clamped frame -- the outputs of
 *    atmvfi_final_residual. */
int atmvfi_conv3x3_planes_readout(const void* in_hi, const void* in_lo, int64_t in_rows, int N, int H, int W, int Cin, const void* w_hi,
                                  const void* w_lo, int Cout, const float* bias, const float* prelu, const void* w2, float* contrib,
                                  int64_t contrib_plane, void* stream);
int atmvfi_refine_tail(const float* contrib, int64_t contrib_plane, const float* bias /*[3] or NULL*/, const float* slope /*[3] or NULL*/,
                       const float* it /*[B,3,H,W]*/, float* it_sum, float* it_clamped, int B, int H, int W, void* stream);

/* Host-boundary frame formats (demo_2x.py:64-85 inference_2frame + benchmark/utils.py:57-80 InputPadder), SURVEY 8f-2.
 *   u8_to_f32: uint8 [H,W,3] (BGR if `bgr`, as cv2 delivers) -> fp32 planar RGB [3,Hp,Wp] = x / 255 with replicate padding,
 *              the source frame sitting at (pad_top, pad_left) of the padded canvas;
 *   f32_to_u8: the inverse: crop, np.round(x * 255) (half to even), uint8 [H,W,3].  Bit-exact against the numpy path. */
int atmvfi_frame_u8_to_f32(const void* src, int H, int W, int bgr, float* dst, int Hp, int Wp, int pad_top, int pad_left, void* stream);
int atmvfi_frame_f32_to_u8(const float* src, int Hp, int Wp, int pad_top, int pad_left, void* dst, int H, int W, int bgr, void* stream);

/* mean |a - b| per sample: global_alignmentness (network_base.py:560-561).  Two passes with a fixed summation order -- the result is
 * run-to-run bit-identical (the ensemble's pick compares these means) -- through `workspace`: at least
 * atmvfi_l1_mean_workspace_floats(B, per_sample) floats of scratch, the caller's. */
int64_t atmvfi_l1_mean_workspace_floats(int B, int64_t per_sample);
int atmvfi_l1_mean(const float* a, const float* b, float* out, int B, int64_t per_sample, float* workspace, int64_t workspace_floats, void* stream);
/* multiscale_global_motion_ensemble's per-sample pick (network_base.py:591-603): out0 / out1 [B, per_sample] = the candidate flow pair
 * (c0_lL, c1_lL, already at the level-0 flow resolution) of the level whose loss[b] is the minimum, the first one on ties (the
 * reference's min() / if / elif chain, NaN losses included: a NaN loss of level 0 falls through to level 2).  Keeps the ensemble forward free of device arithmetic outside this ABI. */
int atmvfi_ensemble_select(const float* loss0, const float* loss1, const float* loss2, const float* c0_l0, const float* c1_l0,
                           const float* c0_l1, const float* c1_l1, const float* c0_l2, const float* c1_l2, float* out0, float* out1, int B,
                           int64_t per_sample, void* stream);

/* The encoder's full-resolution stem in one launch: feat_extracts.0.0 (3 -> C0, 3x3 + PReLU), feat_extracts.0.1 (C0 -> C0, 3x3 + PReLU) and
 * feat_extracts.1.0 (C0 -> C1, 3x3 stride 2 + PReLU) of shared_feat_extraction (network_base.py:99-110, 342-352), (C0, C1) = (24, 48) or
 * (16, 32).  x: the NHWC4-packed frames [F, H, W, 4] (atmvfi_pack_frames), H and W even.  The two full-resolution C0-channel maps never
 * leave the chip (LDS); the result is written as split planes (chunk-major, out_plane_rows >= F * H/2 * W/2 rows per 32-channel chunk),
 * rows = pixels (f, y, x) of the half-resolution map.  All three layers use the f16x3 split (x = hi + lo'/1024, three 16-bit MFMAs per
 * product, fp32 accumulation).  Weights as the host packs them (atm-vfi_amd/hip_ops.py::HipOps.pack_stem), fp16 hi / lo' planes:
 *   w1 [round_up(C0, 16)][32], k = (ky * 3 + kx) * 3 + ci (27 values, then zeros);  w2 [NS][round_up(C0, 16)][32] and w3 [NS][C1][32],
 *   NS = ceil(9 C0 / 32) k-steps of 32 k-values, k = (ky * 3 + kx) * C0 + c, zero beyond 9 C0 and in padded rows;
 *   b1 / p1 and b2 / p2 [round_up(C0, 16)] (bias 0, slope 1 in the padding), b3 / p3 [C1]: bias and PReLU slope of each layer. */
int atmvfi_stem_fused(const float* x, int F, int H, int W, int C0, int C1, const void* w1_hi, const void* w1_lo, const float* b1, const float* p1,
                      const void* w2_hi, const void* w2_lo, const float* b2, const float* p2, const void* w3_hi, const void* w3_lo,
                      const float* b3, const float* p3, void* out_hi, void* out_lo, int64_t out_plane_rows, void* stream);

/* ------------------------------------------------------------------------------------
 * Launch plans: one forward of the hot path as ONE call.
 *
 * The reference's callers run `model(im0, im1)` in a loop on frames of one size (benchmark/test_vimeo90k.py: 3 782 triplets of
 * 256x448; demo_2x.py:129-168: every pair of a video).  For such a loop the sequence of launches of a forward is fixed: same entry
 * points, same arguments, except the pointers into the caller's two frames and into the freshly allocated output tensors.  The host
 * records that sequence once (every launch entry point above, arguments in declaration order, stream excluded) and replays it:
 * atmvfi_plan_run() first patches the arguments that point into per-call memory -- ops[p.op].a[p.arg] = slots[p.slot] + p.offset --
 * and then issues the ops in order on `stream`, stopping at the first failure (*failed_op = its index; the return value and
 * atmvfi_last_error() are that op's).  Nothing is cached inside the library: the plan is the caller's array.
 * atmvfi_plan_fn_id(name) gives the `fn` of an entry point by name (-1 if it is not a launch entry point).
 *
 * LANES (round 5; small frames: a forward at 256x256 is ~120 launches of 5-18 us on a few dozen CUs each, and its dependency graph has
 * independent branches -- the local cross-scale fusion beside the global branch's feature extraction, the two feature-enhancement blocks
 * beside the local motion MLP): atmvfi_plan_run_lanes() issues op i on streams[lanes[i]] and understands two more ops,
 *   fn = ATMVFI_PLAN_RECORD, a[0].i = e : hipEventRecord(events[e], streams[lanes[i]])
 *   fn = ATMVFI_PLAN_WAIT,   a[0].i = e : hipStreamWaitEvent(streams[lanes[i]], events[e])
 * The streams and events are the caller's (lane 0 = the stream whose order the caller sees; every other lane must be joined back into
 * it -- RECORD on that lane, WAIT on lane 0 -- before the plan ends); the same kernels run, in a partial order: results are bit-identical
 * to atmvfi_plan_run() on one stream as long as concurrent branches touch disjoint memory, which is the recorder's obligation.
 * ---------------------------------------------------------------------------------- */
typedef union atmvfi_plan_arg { uint64_t u; int64_t i; double f; } atmvfi_plan_arg;
#define ATMVFI_PLAN_MAX_ARGS 28
typedef struct atmvfi_plan_op {
    int32_t fn;       /* atmvfi_plan_fn_id() */
    int32_t nargs;    /* must equal the entry point's argument count without the stream (checked) */
    atmvfi_plan_arg a[ATMVFI_PLAN_MAX_ARGS];   /* pointers and sizes as .u / .i (64-bit), float arguments as .f */
} atmvfi_plan_op;
typedef struct atmvfi_plan_patch { int32_t op, arg, slot, reserved; int64_t offset; } atmvfi_plan_patch;
int atmvfi_plan_fn_id(const char* name);
int atmvfi_plan_run(atmvfi_plan_op* ops, int n_ops, const atmvfi_plan_patch* patches, int n_patches, const uint64_t* slots, int n_slots,
                    int* failed_op, void* stream);
#define ATMVFI_PLAN_RECORD (-2)
#define ATMVFI_PLAN_WAIT (-3)
int atmvfi_plan_run_lanes(atmvfi_plan_op* ops, int n_ops, const int32_t* lanes, const atmvfi_plan_patch* patches, int n_patches,
                          const uint64_t* slots, int n_slots, int* failed_op, void* const* streams, int n_streams, void* const* events,
                          int n_events);

#ifdef __cplusplus
}
#endif
#endif /* ATMVFI_H */

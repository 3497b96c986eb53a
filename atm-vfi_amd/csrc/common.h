// Shared helpers of libatmvfi_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <type_traits>

#include <atomic>

#include "atmvfi.h"

namespace atmvfi {

void set_error(const char* fmt, ...);
int check_launch(const char* what);

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
constexpr int kMaxDevices = 64;
// compute units of the CURRENT device rounded down to a multiple of 8 (persistent grids keep block % 8 == XCD group); cached per
// device id (one process may drive several GPUs)
static inline int cu_count() {
    static std::atomic<int> cache[kMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 256;
    int n = cache[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
        n = cus / 8 * 8;
        cache[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}
// hipFuncAttributeMaxDynamicSharedMemorySize of one kernel, raised once per device (the attribute belongs to the device's copy of
// the function; a process-wide "done" flag would leave the second GPU of a process at the 64 KiB default)
template <auto Kern>
static inline hipError_t allow_dynamic_lds(size_t bytes) {
    static std::atomic<int> granted[kMaxDevices];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const bool cached = dev >= 0 && dev < kMaxDevices;
    if (cached && granted[dev].load(std::memory_order_acquire) >= (int)bytes) return hipSuccess;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(Kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && cached) granted[dev].store((int)bytes, std::memory_order_release);
    return e;
}

}  // namespace atmvfi

#define ATMVFI_REQUIRE(cond, code, ...)          \
    do {                                         \
        if (!(cond)) {                           \
            atmvfi::set_error(__VA_ARGS__);      \
            return (code);                       \
        }                                        \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// MODE.FP16_OVFL = 1 for the rest of the wave: a conversion to fp16 that overflows returns +-65504 instead of +-inf (a true
// inf stays inf; tools/probes/f16_ovfl_probe.hip).  Every kernel that splits values calls this first: the split below relies on
// it for its saturation.
__device__ __forceinline__ void fp16_saturate_on() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1"); }

// The hi/lo split of the f16x3 engines on two values at once: hi = fp16(x), lo' = fp16((x - hi) * 1024), both conversions
// saturating at +-65504 through MODE.FP16_OVFL (fp16_saturate_on() at the kernel's top).  Four VALU instructions per two values:
// v_cvt_pk_f16_f32, v_pk_mul_f32 (x * 1024), and v_fma_mixlo_f16 / v_fma_mixhi_f16 computing fp16(fma(hi, -1024, x * 1024)) with hi
// read as the fp16 half it is -- x - hi and both products are exact in fp32, so the one rounding to fp16 sees the same real number as
// the six-instruction form (v_cvt_pk, two v_cvt_f32_f16, v_pk_add_f32, v_pk_mul_f32, v_cvt_pk) it replaces: bit-identical for all 2^32
// fp32 bit patterns under FP16_OVFL = 1 (tools/probes/split_mix_probe.hip, profiles/r04_split_mix_probe.txt).  The epilogues and
// operand stagings that split are VALU-bound.  (Until round 2 the saturation was four v_med3_f32: 5 per value.)  An infinite input
// stays infinite in hi and makes lo' NaN: the fp32 reference has inf or NaN downstream of such a value as well.
#ifdef ATMVFI_RANGE_CHECK
// The CHECKED build (libatmvfi_hip_checked.so, `make checked`; Network.set_precision("f16x3-checked")): every activation that is
// split counts, in a device word the caller attached (atmvfi_range_word_set), when its hi half comes out at the fp16 limit or
// non-finite -- |x| >= 65488 (rounds or saturates to 65504), inf, NaN -- i.e. when the f16x3 engines' operand range contract
// (deviation 2) is violated and the result silently differs from the fp32 reference.  One device variable
// and one setter per translation unit (the library is not linked with relocatable device code); the default build has neither.
namespace atmvfi {
typedef int (*RangeWordSetter)(unsigned*, hipStream_t);
void range_registry_add(RangeWordSetter s);
}  // namespace atmvfi
namespace {
__device__ unsigned* g_range_word = nullptr;
unsigned* g_range_word_host = nullptr;
int range_word_set_tu(unsigned* p, hipStream_t s) {
    g_range_word_host = p;           // (a source that outlives the asynchronous copy)
    return (int)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_range_word), &g_range_word_host, sizeof(unsigned*), 0, hipMemcpyHostToDevice, s);
}
const int g_range_word_registered = (atmvfi::range_registry_add(&range_word_set_tu), 0);
}  // namespace
#endif

__device__ __forceinline__ void split_pair(const f32x2 x, f16x2& hi, f16x2& lo) {
    hi = __builtin_convertvector(x, f16x2);
    const f32x2 xs = x * 1024.0f;
    const unsigned hb = __builtin_bit_cast(unsigned, hi);
#ifdef ATMVFI_RANGE_CHECK
    if ((hb & 0x7fffu) >= 0x7bffu || ((hb >> 16) & 0x7fffu) >= 0x7bffu) {
        unsigned* w = g_range_word;
        if (w) atomicAdd(w, 1u);
    }
#endif
    const float k = 1024.0f;
    unsigned lb;
    asm("v_fma_mixlo_f16 %0, %1, -%2, %3 op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(hb), "v"(k), "v"(xs.x));
    asm("v_fma_mixhi_f16 %0, %1, -%2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lb) : "v"(hb), "v"(k), "v"(xs.y));
    lo = __builtin_bit_cast(f16x2, lb);
}

// Output of a row-producing kernel: fp32 rows, and/or the split-plane pair (hi = fp16(x), lo = fp16((x - hi) * 1024), both
// saturating) that the split GEMM (gemm_split.hip) reads by LDS-DMA.  Either side may be null.
// Plane layout, CHUNK MAJOR: element (row, c) lives at ((c / 32) * plane_rows + row) * 32 + c % 32, i.e. [32-channel chunk][row][32].
// The 16 rows x 64 bytes that one LDS-DMA instruction of the GEMM moves are then one contiguous KiB; with row-major planes they
// were 16 pieces a row pitch apart, 2.6x dearer to issue (tools/probes/dma_probe.hip).
struct RowSink {
    float* f32;
    long long ld;
    _Float16* hi;
    _Float16* lo;
    long long plane_rows;
};
__device__ __forceinline__ void sink_store4(const RowSink& s, long long row, int c, const f32x4 v) {
    if (s.f32) *reinterpret_cast<f32x4*>(s.f32 + row * s.ld + c) = v;
    if (s.hi) {
        f16x2 h0, l0, h1, l1;
        split_pair((f32x2){v.x, v.y}, h0, l0);
        split_pair((f32x2){v.z, v.w}, h1, l1);
        const f16x4 h = {h0.x, h0.y, h1.x, h1.y}, l = {l0.x, l0.y, l1.x, l1.y};
        const long long off = ((long long)(c >> 5) * s.plane_rows + row) * 32 + (c & 31);
        *reinterpret_cast<f16x4*>(s.hi + off) = h;
        *reinterpret_cast<f16x4*>(s.lo + off) = l;
    }
}
static inline bool sink_ok(const void* f32, int ld, int C, const void* hi, const void* lo, long long plane_rows, long long rows) {
    if (!f32 && !hi) return false;
    if (f32 && (ld % 4 != 0 || ld < C || (reinterpret_cast<uintptr_t>(f32) & 15u))) return false;
    if ((hi == nullptr) != (lo == nullptr)) return false;
    if (hi && (plane_rows < rows || (reinterpret_cast<uintptr_t>(hi) & 15u) || (reinterpret_cast<uintptr_t>(lo) & 15u))) return false;
    return true;
}

// Per-column epilogue constants by LDS-DMA (used by the f16x3 engines; rationale in gemm_common.h): every wave issues exactly ONE
// 4-byte-per-lane global_load_lds, piece (wave mod NPIECE) of the round_up(2*BN, 64) floats cst[c] = bias[co(c)],
// cst[BN + c] = slope[co(c)], with 0 / 1 for absent arrays or columns that map to no channel (co < 0).
namespace {
__device__ const float kEpilogueDefaults[2] = {0.f, 1.f};
}
constexpr int epilogue_const_floats(int BN) { return (2 * BN + 63) / 64 * 64; }
template <int BN, typename ColToChannel>
__device__ __forceinline__ void dma_epilogue_consts(const float* bias, const float* slope, int n0, float* cst, int wave, int lane,
                                                    ColToChannel channel_of) {
    constexpr int NPIECE = (2 * BN + 63) / 64;
    const int piece = wave % NPIECE;
    const int t = piece * 64 + lane;
    const bool sl = t >= BN;
    const int co = t < 2 * BN ? channel_of(n0 + (sl ? t - BN : t)) : -1;
    const float* src = sl ? slope : bias;
    const float* p = (src && co >= 0) ? src + co : &kEpilogueDefaults[sl ? 1 : 0];
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                     (__attribute__((address_space(3))) void*)(cst + piece * 64), 4, 0, 0);
}

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
// ---- fragment reads under manual wait counts (conv3x3_f16x3_row.hip explains why; also gemm_split.hip) ----
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
// 32-bit LDS offset of a pointer into __shared__ memory (the low half of the flat address)
__device__ __forceinline__ unsigned lds_offset(const void* p) { return (unsigned)(unsigned long long)p; }
template <int OFF>
__device__ __forceinline__ void lds_read16(f16x8_t& d, unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");     // "memory": stays behind the barrier / ds_writes that publish the data
}
// s_waitcnt lgkmcnt(N), tied to the registers it makes valid so that their consumers cannot be scheduled above it
template <int N>
__device__ __forceinline__ void lds_wait2(f16x8_t& a, f16x8_t& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int N>
__device__ __forceinline__ void lds_wait4(f16x8_t& a, f16x8_t& b, f16x8_t& c, f16x8_t& d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}

// exact GELU (erf form), as nn.GELU() default
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// Branch-free erf for the dw-conv + GELU kernel, which the library erff bound (a divergent |z| < 1 branch that most waves take
// both ways, ~45 VALU instructions and ~390 register moves per 32 values).  Two Chebyshev fits evaluated for every lane and
// selected: |z| < 1: z * P5(z^2);  1 <= |z| <= 4: 1 - 2^P7(|z|) with P7 ~ log2(erfc) (v_exp_f32);  beyond 4 erf is 1 in fp32.
// Evaluated in fp32 (tools/fit_erf.py): |erf error| <= 1.2e-7, GELU within 9.2e-8 of the exact one -- the same as
// 0.5 x (1 + erff(x / sqrt 2)) with a correctly rounded erff (8.3e-8).
__device__ __forceinline__ float erf_2range(float z) {
    const float az = fabsf(z);
    const float u = az * az;
    float pa = fmaf(-5.654105859e-04f, u, 4.923277665e-03f);
    pa = fmaf(pa, u, -2.671638510e-02f);
    pa = fmaf(pa, u, 1.128036441e-01f);
    pa = fmaf(pa, u, -3.761234978e-01f);
    pa = fmaf(pa, u, 1.128379127e+00f);
    const float ea = az * pa;
    const float ac = fminf(az, 4.0f);
    float pb = fmaf(-1.920139151e-05f, ac, 4.581595994e-04f);
    pb = fmaf(pb, ac, -4.958351839e-03f);
    pb = fmaf(pb, ac, 3.263662691e-02f);
    pb = fmaf(pb, ac, -1.490577801e-01f);
    pb = fmaf(pb, ac, -9.220167627e-01f);
    pb = fmaf(pb, ac, -1.624367783e+00f);
    pb = fmaf(pb, ac, -1.092074884e-03f);
    const float eb = 1.0f - __builtin_amdgcn_exp2f(pb);
    return copysignf(az < 1.0f ? ea : eb, z);
}
__device__ __forceinline__ float gelu_erf2(float x) { return 0.5f * x * (1.0f + erf_2range(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

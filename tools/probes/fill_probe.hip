// Sustained global -> LDS fill rate of one CU on gfx950 (the operand stream of the LDS-DMA GEMMs): every wave of a 512- or 256-thread
// workgroup keeps DEPTH one-KiB global_load_lds_dwordx4 pieces in flight (counted s_waitcnt vmcnt) and streams a source region of a chosen
// size round and round, one workgroup per CU (or two).  Regions: 3 MiB read by every workgroup or 3 MiB per XCD (L2 hits), 128 MiB (Infinity Cache), 2 GiB
// (HBM).  A second mode loads the same pieces into registers (global_load_dwordx4) for comparison.  Prints bytes / shader clock / CU and
// GB/s per CU and chip-wide.
//
//   hipcc -O3 --offload-arch=gfx950 tools/probes/fill_probe.hip -o tools/probes/fill_probe && tools/probes/fill_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(const unsigned char* src, unsigned char* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
// each wave walks its own sequence of KiB pieces: piece index = (it * nwaves_total + global wave) mod pieces_in_region, so that at any
// moment the CUs of the chip read different lines (SHARED = 0) or every workgroup reads the same sequence (SHARED = 1: a weight stream)
template <int DEPTH, bool TO_REG, int SHARED>
__global__ void fill(const unsigned char* src, unsigned long long region_pieces, int iters, unsigned long long* out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    // SHARED 0: every wave of the chip its own pieces of the region; 1: every workgroup the same sequence; 2: the region is cut into 8
    // parts, one per XCD (workgroup b runs on XCD b & 7), the waves of an XCD share their part
    const unsigned long long gw = SHARED == 1 ? (unsigned long long)wave
                                : SHARED == 2 ? (unsigned long long)(blockIdx.x >> 3) * nw + wave : (unsigned long long)blockIdx.x * nw + wave;
    const unsigned long long stride = SHARED == 1 ? (unsigned long long)nw
                                    : SHARED == 2 ? (unsigned long long)(gridDim.x >> 3) * nw : (unsigned long long)gridDim.x * nw;
    if (SHARED == 2) { region_pieces >>= 3; src += (unsigned long long)(blockIdx.x & 7) * region_pieces * 1024; }
    unsigned char* dst = smem + wave * DEPTH * 1024;
    f32x4 r[DEPTH];
    for (int d = 0; d < DEPTH; ++d) r[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long piece = gw % region_pieces;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const unsigned char* p = src + piece * 1024 + lane * 16;
            if constexpr (TO_REG) {
                // the load issued DEPTH pieces ago into r[d] has landed (at most DEPTH - 1 younger ones stay in flight); only then is
                // r[d] the target of the next one -- an asm load writes its register asynchronously, so the register must not be
                // anything the compiler may reuse before the wait (a temporary here once became the next address: a GPU fault)
                asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r[d]) : "n"(DEPTH - 1) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(r[d]) : "v"(p) : "memory");
            } else {
                dma16(p, dst + d * 1024);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
            }
            piece += stride;
            if (piece >= region_pieces) piece -= region_pieces;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (TO_REG) { for (int d = 0; d < DEPTH; ++d) asm volatile("" : "+v"(r[d])); }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long w1 = __builtin_amdgcn_s_memrealtime();
    if (TO_REG) { float s = 0; for (int d = 0; d < DEPTH; ++d) s += r[d].x; if (s == 12345.f) sink[0] = s; }
    if (lane == 0) { out[((unsigned long long)blockIdx.x * nw + wave) * 2] = t1 - t0; out[((unsigned long long)blockIdx.x * nw + wave) * 2 + 1] = w1 - w0; }
}
template <int DEPTH, bool TO_REG, int SHARED>
static void run(const char* what, const unsigned char* src, size_t region, int wgs_per_cu, int threads, unsigned long long* out, float* sink) {
    const int cus = 256, grid = cus * wgs_per_cu, nw = threads / 64, iters = 4000 / DEPTH;
    const size_t lds = TO_REG ? 64 : (size_t)nw * DEPTH * 1024;
    hipFuncSetAttribute((const void*)fill<DEPTH, TO_REG, SHARED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((fill<DEPTH, TO_REG, SHARED>), dim3(grid), dim3(threads), lds, 0, src, (unsigned long long)(region / 1024), iters, out, sink);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h((size_t)grid * nw * 2);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double clk = 0, wall = 0;
    for (size_t i = 0; i < h.size(); i += 2) { clk += h[i]; wall += h[i + 1]; }
    clk /= h.size() / 2; wall /= h.size() / 2;                        // per wave: shader clocks, 100 MHz ticks
    const double bytes_cu = (double)wgs_per_cu * nw * iters * DEPTH * 1024;
    const double secs = wall / 100e6;
    printf("%-10s %s region %7.1f MiB, %d x %d threads per CU, %2d KiB in flight per wave: %5.1f B/clk/CU, %6.1f GB/s per CU, %5.2f TB/s chip (%.0f MHz)\n",
           TO_REG ? "registers" : "LDS-DMA", what, region / 1048576.0, wgs_per_cu, threads, DEPTH, bytes_cu / clk, bytes_cu / secs / 1e9,
           bytes_cu * cus / secs / 1e12, clk / wall * 100.0);
}
int main() {
    unsigned char* src; unsigned long long* out; float* sink;
    const size_t n = (size_t)2 << 30;
    hipMalloc(&src, n); hipMemset(src, 1, n); hipMalloc(&out, 256 * 2 * 16 * 2 * 8); hipMalloc(&sink, 4);
    // (a) a weight-like stream: every workgroup reads the same 3 MiB round and round (L2 hits on every XCD)
    run<6, false, 1>("SHARED", src, (size_t)3 << 20, 1, 512, out, sink);
    run<6, false, 1>("SHARED", src, (size_t)3 << 20, 2, 256, out, sink);
    run<6, false, 2>("XCD   ", src, (size_t)24 << 20, 1, 512, out, sink);
    run<12, false, 2>("XCD   ", src, (size_t)24 << 20, 1, 512, out, sink);
    run<3, false, 2>("XCD   ", src, (size_t)24 << 20, 1, 512, out, sink);
    run<6, false, 2>("XCD   ", src, (size_t)24 << 20, 2, 256, out, sink);
    run<6, false, 2>("XCD   ", src, (size_t)24 << 20, 2, 512, out, sink);
    // (b) activation-like: every wave its own pieces of a region that fits the 8 L2s (24 MiB), the Infinity Cache (128 MiB), HBM (2 GiB)
    run<6, false, 0>("own   ", src, (size_t)128 << 20, 1, 512, out, sink);
    run<6, false, 0>("own   ", src, n, 1, 512, out, sink);
    run<12, false, 0>("own   ", src, n, 1, 512, out, sink);
    run<6, false, 0>("own   ", src, n, 2, 256, out, sink);
    // (c) the same pieces into registers
    run<6, true, 1>("SHARED", src, (size_t)3 << 20, 1, 512, out, sink);
    run<6, true, 2>("XCD   ", src, (size_t)24 << 20, 1, 512, out, sink);
    run<6, true, 0>("own   ", src, n, 1, 512, out, sink);
    return 0;
}

// LDS-DMA issue probe on gfx950: per-wave ticks to issue 6 global_load_lds_dwordx4 (a) alone, (b) followed by s_waitcnt lgkmcnt(0),
// (c) followed by s_waitcnt vmcnt(0); 256 workgroups x 512 threads, sources streaming through L2 (each piece = 16 rows x 64 B or
// one contiguous KiB).  Answers: does an in-flight LDS-DMA hold lgkmcnt (i.e. does the next ds_read wait for it)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h16;
__device__ __forceinline__ void dma16(const h16* src, h16* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}
template <int MODE, int CONTIG>
__global__ __launch_bounds__(512) void probe(const h16* src, long long stride_rows, unsigned long long* out, int iters, int shared_src) {
    extern __shared__ __attribute__((aligned(16))) h16 smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long acc = 0;
    // shared_src: every workgroup streams the SAME addresses (weights of a convolution), else each wave its own
    const h16* base = src + ((long long)(shared_src ? 0 : blockIdx.x) * 8 + wave) * 6 * 16 * (CONTIG ? 32 : stride_rows);
    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_s_barrier();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const h16* p = CONTIG ? base + (k * 16 + (lane >> 2)) * 32 + (lane & 3) * 8
                                  : base + (long long)(k * 16 + (lane >> 2)) * stride_rows + (lane & 3) * 8;
            dma16(p + (long long)it * (shared_src ? 8 * 6 * 16 * 32 : 64 * 4096), smem + (wave * 6 + k) * 512);
        }
        if (MODE == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        acc += t1 - t0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (lane == 0) out[blockIdx.x * 8 + wave] = acc;
}
int main() {
    h16* src; unsigned long long* out;
    const size_t n = (size_t)1 << 30;           // 2 GiB of halves
    hipMalloc(&src, n * 2); hipMemset(src, 0, n * 2); hipMalloc(&out, 2048 * 8);
    const int iters = 50;
    for (int shared = 0; shared < 2; ++shared)
    for (int contig = 0; contig < 2; ++contig)
        for (int mode = 0; mode < 3; mode += 2) {
            for (int rep = 0; rep < 2; ++rep) {
#define L(M, C) hipLaunchKernelGGL((probe<M, C>), dim3(256), dim3(512), 8 * 6 * 1024, 0, src, 4096LL, out, iters, shared)
                if (contig == 0) { if (mode == 0) L(0, 0); else if (mode == 1) L(1, 0); else L(2, 0); }
                else { if (mode == 0) L(0, 1); else if (mode == 1) L(1, 1); else L(2, 1); }
                hipDeviceSynchronize();
            }
            unsigned long long h[2048]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
            double m = 0; for (auto v : h) m += v; m /= 2048 * iters;
            printf("%s%s source, %s: %.0f ticks per 6 DMA instructions per wave\n", shared ? "SHARED " : "", contig ? "contiguous-KiB" : "16-row-scattered",
                   mode == 0 ? "issue only" : mode == 1 ? "issue + lgkmcnt(0)" : "issue + vmcnt(0)", m);
        }
    return 0;
}

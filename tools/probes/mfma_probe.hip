// Bare-MFMA probes on gfx950: cycles per v_mfma_f32_16x16x32_f16 with 1 or 2 waves per SIMD, operands in registers,
// 16 or 32 independent accumulators, random data.  Build: hipcc -O3 --offload-arch=gfx950 mfma_probe.hip -o mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(512, 2) void probe(const _Float16* src, float* out, unsigned long long* cyc, int iters) {
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = *reinterpret_cast<const f16x8*>(src + (threadIdx.x * 8 + i * 4096) % 65536);
        b[i] = *reinterpret_cast<const f16x8*>(src + (threadIdx.x * 8 + i * 4096 + 2048) % 65536);
    }
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f32x4 s = (f32x4){0, 0, 0, 0};
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s.z + s.w;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

int main() {
    std::vector<_Float16> h(65536 + 64);
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    _Float16* d; float* o; unsigned long long* c;
    hipMalloc(&d, h.size() * 2); hipMalloc(&o, 1024 * 512 * 4); hipMalloc(&c, 1024 * 8 * 8);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int iters = 2000;
    for (int threads : {256, 512}) {
        for (int nacc : {16, 32}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (nacc == 16) hipLaunchKernelGGL(probe<16>, dim3(256), dim3(threads), 0, 0, d, o, c, iters);
                else hipLaunchKernelGGL(probe<32>, dim3(256), dim3(threads), 0, 0, d, o, c, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> hc(256 * threads / 64);
            hipMemcpy(hc.data(), c, hc.size() * 8, hipMemcpyDeviceToHost);
            double mean = 0; for (auto v : hc) mean += v; mean /= hc.size();
            const double nm = (double)iters * nacc;
            const double waves_per_simd = threads / 256.0;
            printf("threads %d (%.0f wave/SIMD) acc %d: %.1f cycles per MFMA per wave -> %.1f per SIMD slot; %.3f ms -> %.0f TF/s fp16\n", threads,
                   waves_per_simd, nacc, mean / nm, mean / nm / waves_per_simd, ms,
                   256.0 * threads / 64 * nm * 16384 / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}

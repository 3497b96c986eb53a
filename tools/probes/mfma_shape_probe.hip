// Sustained rate of the two fp16 MFMA shapes on gfx950 with the whole chip loaded for tens of milliseconds (the power-managed clock, not
// the boost clock, is what a long kernel gets): v_mfma_f32_16x16x32_f16 (16 cycles, 16 Ki flop) against v_mfma_f32_32x32x16_f16 (32 cycles,
// 32 Ki flop: the same flop per cycle, a quarter of the operand-register reads per flop).  Operands in registers, random data.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_shape_probe.hip -o tools/probes/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>      // 0: 16x16x32 with 32 accumulators of 4 registers; 1: 32x32x16 with 8 accumulators of 16 registers (128 registers each way)
__global__ __launch_bounds__(512, 2) void probe(const _Float16* src, float* out, unsigned long long* cyc, int iters) {
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = *reinterpret_cast<const f16x8*>(src + (threadIdx.x * 8 + i * 4096) % 65536);
        b[i] = *reinterpret_cast<const f16x8*>(src + (threadIdx.x * 8 + i * 4096 + 2048) % 65536);
    }
    float total = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
    if constexpr (SHAPE == 0) {
        f32x4 acc[32];
        for (int i = 0; i < 32; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 32; ++i) total += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    } else {
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[(i >> 2) & 1], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) for (int k = 0; k < 16; ++k) total += acc[i][k];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long w1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = total;
    if ((threadIdx.x & 63) == 0) { cyc[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2] = t1 - t0; cyc[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2 + 1] = w1 - w0; }
}

int main() {
    std::vector<_Float16> h(65536 + 64);
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    _Float16* d; float* o; unsigned long long* c;
    if (hipMalloc(&d, h.size() * 2) != hipSuccess || hipMalloc(&o, 512 * 512 * 4) != hipSuccess || hipMalloc(&c, 512 * 8 * 16) != hipSuccess) return 1;
    if (hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice) != hipSuccess) return 1;
    for (int rep = 0; rep < 2; ++rep)
    for (int shape = 0; shape < 2; ++shape)
        for (int threads : {256, 512}) {
            const int per_iter = shape == 0 ? 32 : 8;
            const double flop = shape == 0 ? 16384.0 : 32768.0;
            const int iters = 60000 * (shape == 0 ? 1 : 2) / (threads / 256);             // ~50-60 ms each
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            if (shape == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(threads), 0, 0, d, o, c, iters);
            else hipLaunchKernelGGL(probe<1>, dim3(256), dim3(threads), 0, 0, d, o, c, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> hc(256 * threads / 64 * 2);
            (void)hipMemcpy(hc.data(), c, hc.size() * 8, hipMemcpyDeviceToHost);
            double clk = 0, wall = 0;
            for (size_t i = 0; i < hc.size(); i += 2) { clk += hc[i]; wall += hc[i + 1]; }
            const double nm = (double)iters * per_iter;
            printf("%s, %d wave(s) per SIMD: %.1f shader clocks per MFMA per SIMD, clock %.0f MHz, %.1f ms -> %.0f TFLOP/s fp16\n",
                   shape == 0 ? "16x16x32" : "32x32x16", threads / 256, clk / (hc.size() / 2) / nm / (threads / 256.0), clk / wall * 100.0, ms,
                   256.0 * threads / 64 * nm * flop / (ms * 1e-3) / 1e12);
        }
    return 0;
}

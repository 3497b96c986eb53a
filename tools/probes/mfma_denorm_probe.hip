// Does v_mfma_f32_16x16x32_f16 honour fp16 SUBNORMAL inputs on gfx950, or flush them to zero?  (Question behind a single-accumulator
// form of the f16x3 split: x = hi + lo with lo UNSCALED is subnormal in fp16 for |x| < 0.125.)  A = one subnormal value a in every
// element, B = 1.0: every output should be 32 * a.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_denorm_probe.hip -o tools/probes/mfma_denorm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, float aval, float bval) {
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)aval; b[e] = (_Float16)bval; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c.x; out[1] = (float)a[0]; }
}
int main() {
    float* d; hipMalloc(&d, 8);
    const float vals[] = {1.0f, 6.2e-5f /* just normal */, 3.0e-5f /* subnormal */, 1.0e-6f, 6.0e-8f /* smallest subnormal */};
    for (float v : vals) {
        float h[2];
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, v, 1.0f);
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("a = %-10.3e (as fp16 %.6e)  sum_k a*1 = %.6e  expected %.6e  %s\n", v, h[1], h[0], 32.0 * h[1], h[0] == 32.0f * h[1] ? "kept" : (h[0] == 0.f ? "FLUSHED" : "other"));
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 1.0f, v);
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("      as the B operand:                       %.6e\n", h[0]);
    }
    return 0;
}

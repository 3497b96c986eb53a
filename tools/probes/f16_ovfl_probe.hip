// Does MODE.FP16_OVFL (hwreg MODE bit 23) make v_cvt_pk_f16_f32 / v_cvt_f16_f32 saturate at +-65504 instead of +-inf on gfx950?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/f16_ovfl_probe.hip -o /tmp/f16_ovfl && /tmp/f16_ovfl
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, float* out, int n, int ovfl) {
    if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
    const int i = threadIdx.x;
    if (i < n) {
        const f32x2 x = {in[2 * i], in[2 * i + 1]};
        const f16x2 h = __builtin_convertvector(x, f16x2);
        _Float16 s = (_Float16)in[2 * i];
        asm volatile("" : "+v"(s));
        out[3 * i] = (float)h.x;
        out[3 * i + 1] = (float)h.y;
        out[3 * i + 2] = (float)s;
    }
    if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 0");
}
int main() {
    const int n = 6;
    float h[2 * n] = {1.0f, -2.5f, 65504.0f, 65520.0f, 1e6f, -1e6f, 7e4f, -7e4f, INFINITY, -INFINITY, NAN, 65519.0f};
    float *d, *o, r[3 * n];
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int ov = 0; ov < 2; ++ov) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n, ov);
        hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
        printf("FP16_OVFL=%d\n", ov);
        for (int i = 0; i < n; ++i) printf("  (%g, %g) -> pk (%g, %g), scalar %g\n", h[2 * i], h[2 * i + 1], r[3 * i], r[3 * i + 1], r[3 * i + 2]);
    }
    return 0;
}

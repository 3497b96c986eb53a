// What bounds dwconv_gelu_rows_kernel (pointwise.hip; 0.94 ms per 1080p forward at 4.2 TB/s of its bytes)?  The product kernel's body in
// variants that drop one thing each (timing only; the variants' results are wrong on purpose):
//   0 as is | 1 no x-neighbour loads (one load per row step: a third of the L1 traffic) | 2 no GELU | 3 = 1 + 2 | 4 split-copy (no conv)
//   5 as is, 32 x-positions x 32 channels per block (half the channel span per pixel: 128-byte pieces)
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I atm-vfi_amd/csrc -I include tools/probes/dwconv_probe.hip -o tools/probes/dwconv_probe
#include "common.h"
#include <cstdio>
#include <vector>

template <int RS, int V, int OCC = 1>
__global__ __launch_bounds__(256, OCC) void dw(const float* __restrict__ in, int in_ld, const RowSink out, const float* __restrict__ w9,
                                          const float* __restrict__ bias, int N, int H, int W, int C, int xblocks, int cblocks, int strips) {
    fp16_saturate_on();
    int bid = blockIdx.x;
    const int cb = bid % cblocks; bid /= cblocks;
    const int xb = bid % xblocks; bid /= xblocks;
    const int sb = bid % strips;
    const int n = bid / strips;
    const int c = (cb * 16 + (threadIdx.x & 15)) << 2;
    const int x = xb * 16 + (threadIdx.x >> 4);
    if (x >= W) return;
    const int y0 = sb * RS;
    f32x4 wv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const f32x4*>(w9 + t * C + c);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + c);
    const bool xm_ok = x > 0, xp_ok = x + 1 < W;
    const float* base = in + ((long long)n * H * W + x) * in_ld + c;
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 win[3][3];
    auto load_row = [&](int y, f32x4 (&dst)[3]) {
        const bool ok = (unsigned)y < (unsigned)H;
        const float* p = base + (long long)(ok ? y : 0) * W * in_ld;
        const f32x4 b = *reinterpret_cast<const f32x4*>(p);
        f32x4 a = b, d = b;
        if (V != 1 && V != 3 && V != 4) {
            a = *reinterpret_cast<const f32x4*>(p - (xm_ok ? in_ld : 0));
            d = *reinterpret_cast<const f32x4*>(p + (xp_ok ? in_ld : 0));
        }
        dst[0] = (ok && xm_ok) ? a : zero;
        dst[1] = ok ? b : zero;
        dst[2] = (ok && xp_ok) ? d : zero;
    };
    load_row(y0 - 1, win[0]);
    load_row(y0, win[1]);
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        const int y = y0 + r;
        load_row(y + 1, win[(r + 2) % 3]);
        if (y < H) {
            f32x4 o;
            if (V == 4) {
                o = win[(r + 1) % 3][1];
            } else {
                f32x2 a01 = bv.xy, a23 = bv.zw;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const f32x4* row = win[(r + ky) % 3];
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const f32x4 v = row[kx], wt = wv[ky * 3 + kx];
                        a01 = __builtin_elementwise_fma(v.xy, wt.xy, a01);
                        a23 = __builtin_elementwise_fma(v.zw, wt.zw, a23);
                    }
                }
                const f32x4 acc = {a01.x, a01.y, a23.x, a23.y};
                if (V == 2 || V == 3) o = acc;
                else { o.x = gelu_erf2(acc.x); o.y = gelu_erf2(acc.y); o.z = gelu_erf2(acc.z); o.w = gelu_erf2(acc.w); }
            }
            sink_store4(out, ((long long)n * H + y) * W + x, c, o);
        }
    }
}

int main() {
    const int N = 2, H = 136, W = 240, C = 1536;
    const long long rows = (long long)N * H * W;
    float *in, *w9, *bias;
    _Float16 *hi, *lo;
    hipMalloc(&in, rows * C * 4); hipMalloc(&w9, 9 * C * 4); hipMalloc(&bias, C * 4);
    hipMalloc(&hi, (rows + 1) * C * 2); hipMalloc(&lo, (rows + 1) * C * 2);
    std::vector<float> h(rows * C);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(in, h.data(), rows * C * 4, hipMemcpyHostToDevice);
    hipMemcpy(w9, h.data(), 9 * C * 4, hipMemcpyHostToDevice); hipMemcpy(bias, h.data(), C * 4, hipMemcpyHostToDevice);
    const RowSink sink{nullptr, 0, hi, lo, rows + 1};
    const int xblocks = (W + 15) / 16, cblocks = C / 64, RS = 16, strips = (H + RS - 1) / RS;
    const unsigned blocks = (unsigned)(N * strips * xblocks * cblocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, const char* name) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, in, C, sink, w9, bias, N, H, W, C, xblocks, cblocks, strips);
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, in, C, sink, w9, bias, N, H, W, C, xblocks, cblocks, strips);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("%-44s %7.1f us  %5.2f TB/s of read+write bytes\n", name, ms * 1e3, 8.0 * rows * C / (ms * 1e-3) / 1e12);
    };
    run(dw<16, 0>, "0 as is (16-row strips)");
    run(dw<16, 1>, "1 no x-neighbour loads");
    run(dw<16, 2>, "2 no GELU");
    run(dw<16, 3>, "3 no x-neighbour loads, no GELU");
    run(dw<16, 4>, "4 split-copy (no conv, no GELU)");
    run(dw<16, 0>, "0 as is again");
    run(dw<16, 0, 5>, "0 as is, launch_bounds(256, 5)");
    run(dw<16, 0, 6>, "0 as is, launch_bounds(256, 6)");
    run(dw<16, 0, 8>, "0 as is, launch_bounds(256, 8)");
    run(dw<16, 0>, "0 as is again");
    return 0;
}

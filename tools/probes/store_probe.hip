// Store-issue probe on gfx950: 256 workgroups x 512 threads, each wave issues 16 global_store_dwordx4 per tile in one of three
// lane->address patterns, tiles back to back.  A: MFMA-layout epilogue (16 rows x 64-byte pieces per instruction),
// B: 4 rows x 256 contiguous bytes per instruction, C: 1 KiB contiguous per instruction.  Reports bytes/clk/CU equivalents.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(512) void probe(float* out, int ld, int tiles, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    f32x4 v = (f32x4){(float)lane, 1.f, 2.f, 3.f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles; ++t) {
        // tile = 256 rows x 128 columns of a [rows][ld] matrix; wave owns rows 32*wave..+31
        float* base = out + ((long long)(blockIdx.x * tiles + t) * 256 + 32 * wave) * ld;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float* p;
            if (PAT == 0) {          // k = (i, j): row 16i + r, columns 16j + 4g  (i < 2, j < 8)
                const int i = k >> 3, j = k & 7;
                p = base + (long long)(16 * i + r) * ld + 16 * j + 4 * g;
            } else if (PAT == 1) {   // 2 rows x 512 B per instruction: row 2k + (lane>>5), columns 4*(lane&31)
                p = base + (long long)(2 * k + (lane >> 5)) * ld + 4 * (lane & 31);
            } else if (PAT == 3) {   // 8 rows x 128 B per instruction: k = (i, jp): row 16i + 8h + (r & 7), columns 32jp + 16(r >> 3) + 4g  (i < 2, jp < 4, h < 2)
                const int i = k >> 3, jp = (k >> 1) & 3, h = k & 1;
                p = base + (long long)(16 * i + 8 * h + (r & 7)) * ld + 32 * jp + 16 * (r >> 3) + 4 * g;
            } else if (PAT == 4) {   // 4 rows x 256 B per instruction: row 4k + (lane >> 4), columns 4 * (lane & 15) + 64 * (k >> 3)  (two column halves)
                p = base + (long long)(4 * (k & 7) + (lane >> 4)) * ld + 4 * (lane & 15) + 64 * (k >> 3);
            } else {                 // fully contiguous 1 KiB per instruction (ld ignored)
                p = out + ((long long)(blockIdx.x * tiles + t) * 256 * 128) + (wave * 16 + k) * 256 + lane * 4;
            }
            *reinterpret_cast<f32x4*>(p) = v;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    const int tiles = 12, ld = 1536;
    float* out; unsigned long long* cyc;
    const size_t n = (size_t)256 * tiles * 256 * ld;
    hipMalloc(&out, n * 4); hipMalloc(&cyc, 256 * 8 * 8);
    for (int nwg : {256, 64, 16})
    for (int pat = 0; pat < 5; ++pat) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (pat == 0) hipLaunchKernelGGL(probe<0>, dim3(nwg), dim3(512), 0, 0, out, ld, tiles, cyc);
            if (pat == 1) hipLaunchKernelGGL(probe<1>, dim3(nwg), dim3(512), 0, 0, out, ld, tiles, cyc);
            if (pat == 3) hipLaunchKernelGGL(probe<3>, dim3(nwg), dim3(512), 0, 0, out, ld, tiles, cyc);
            if (pat == 4) hipLaunchKernelGGL(probe<4>, dim3(nwg), dim3(512), 0, 0, out, ld, tiles, cyc);
            if (pat == 2) hipLaunchKernelGGL(probe<2>, dim3(nwg), dim3(512), 0, 0, out, ld, tiles, cyc);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        unsigned long long h[2048]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double mean = 0; for (int i = 0; i < nwg * 8; ++i) mean += h[i]; mean /= nwg * 8;
        const double bytes = (double)nwg * tiles * 256 * 128 * 4;
        printf("%3d workgroups, pattern %d: %.3f ms, %.2f TB/s, %.0f ticks per tile per wave (issue), %.1f B/tick/CU\n", nwg, pat, ms, bytes / ms / 1e9, mean / tiles,
               256.0 * 128 * 4 / (mean / tiles));
    }
    return 0;
}

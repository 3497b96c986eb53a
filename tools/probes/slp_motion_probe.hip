// Reduced repro of the fp32 motion-expectation chain that window_attn_x3_kernel<9> carried until round 2 (the chain that made
// csrc/Makefile build attention.hip with -fno-slp-vectorize): sum_k P[q,k] (k_xy - q_xy) with the key index on the register axis,
// lane-local partial sums, two __shfl_xor.  Same launch shape (9 waves), same index arithmetic, no MFMA, no LDS.
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/probes/slp_motion_probe.hip -o tools/probes/slp_motion_probe_on
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize tools/probes/slp_motion_probe.hip -o tools/probes/slp_motion_probe_off
//
// Each binary runs the kernel on seeded probabilities and compares with the same arithmetic on the host (operation by operation in
// fp32, so a correct build matches bit for bit up to the shuffle order, which is fixed); prints the number of mismatching (q, x|y)
// entries and the first few of them.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(64 * NT) void motion_chain(const float* __restrict__ p_in, float* __restrict__ out, int N, int ws, int items) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, g = lane >> 4;
    const int q = 16 * w + r;
    const bool qok = q < N;
    for (int item = blockIdx.x; item < items; item += gridDim.x) {
        f32x4 s[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) s[kt] = *reinterpret_cast<const f32x4*>(p_in + (((long long)item * 64 * NT + tid) * NT + kt) * 4);
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = 16 * kt + 4 * g + e;
                if (key >= N) s[kt][e] = 0.f;
                sum += s[kt][e];
            }
        const float inv_ws = 1.0f / (float)ws;
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        const float qy = floorf(((float)q + 0.5f) * inv_ws), qx_ = (float)q - qy * (float)ws;
        float mox = 0.f, moy = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = 16 * kt + 4 * g + e;
                const float p = s[kt][e] * inv;
                s[kt][e] = p;
                const float ky = floorf(((float)key + 0.5f) * inv_ws);
                mox += p * (((float)key - ky * (float)ws) - qx_);
                moy += p * (ky - qy);
            }
        mox += __shfl_xor(mox, 16);
        mox += __shfl_xor(mox, 32);
        moy += __shfl_xor(moy, 16);
        moy += __shfl_xor(moy, 32);
        if (g == 0 && qok) {
            float* mp = out + ((long long)item * N + q) * 2;
            mp[0] = mox;
            mp[1] = moy;
        }
        // keep the probabilities alive like the kernel does (they feed the PV product there)
        float keep = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) keep += s[kt][0] + s[kt][1] + s[kt][2] + s[kt][3];
        if (keep == -1.0f) out[0] = keep;
    }
}

int main() {
    constexpr int NT = 9;
    const int N = 144, ws = 12, items = 64, T = 64 * NT;
    std::vector<float> p((size_t)items * T * NT * 4);
    unsigned st = 12345u;
    for (auto& v : p) { st = st * 1664525u + 1013904223u; v = (float)((st >> 8) & 0xffff) / 65536.0f + 1e-3f; }
    float *dp, *dout;
    hipMalloc(&dp, p.size() * 4);
    hipMalloc(&dout, (size_t)items * N * 2 * 4);
    hipMemcpy(dp, p.data(), p.size() * 4, hipMemcpyHostToDevice);
    hipMemset(dout, 0, (size_t)items * N * 2 * 4);
    hipLaunchKernelGGL(motion_chain<NT>, dim3(32), dim3(T), 0, 0, dp, dout, N, ws, items);
    std::vector<float> out((size_t)items * N * 2);
    if (hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 2; }
    // host restatement, same operation order (lane-local sums over kt, e; then + lane^16; then + lane^32)
    int bad = 0;
    const float inv_ws = 1.0f / (float)ws;
    for (int item = 0; item < items; ++item)
        for (int q = 0; q < N; ++q) {
            const int w = q >> 4, r = q & 15;
            float part_s[4], part_x[4], part_y[4];
            auto P = [&](int g, int kt, int e) { const int key = 16 * kt + 4 * g + e; const int tid = 64 * w + 16 * g + r;
                                                 return key >= N ? 0.f : p[(((size_t)item * T + tid) * NT + kt) * 4 + e]; };
            for (int g = 0; g < 4; ++g) { float s = 0.f; for (int kt = 0; kt < NT; ++kt) for (int e = 0; e < 4; ++e) s += P(g, kt, e); part_s[g] = s; }
            // shfl_xor 16 then 32 as seen from lane group 0: (s0 + s1) + (s2 + s3)
            const float sum = (part_s[0] + part_s[1]) + (part_s[2] + part_s[3]);
            const float inv = 1.0f / sum;
            const float qy = floorf(((float)q + 0.5f) * inv_ws), qx_ = (float)q - qy * (float)ws;
            for (int g = 0; g < 4; ++g) {
                float mx = 0.f, my = 0.f;
                for (int kt = 0; kt < NT; ++kt) for (int e = 0; e < 4; ++e) {
                    const int key = 16 * kt + 4 * g + e;
                    const float pp = P(g, kt, e) * inv;
                    const float ky = floorf(((float)key + 0.5f) * inv_ws);
                    mx += pp * (((float)key - ky * (float)ws) - qx_);
                    my += pp * (ky - qy);
                }
                part_x[g] = mx; part_y[g] = my;
            }
            const float mox = (part_x[0] + part_x[1]) + (part_x[2] + part_x[3]);
            const float moy = (part_y[0] + part_y[1]) + (part_y[2] + part_y[3]);
            const float gx = out[((size_t)item * N + q) * 2], gy = out[((size_t)item * N + q) * 2 + 1];
            const bool bx = !(fabsf(gx - mox) <= 1e-5f * (1.f + fabsf(mox))), by = !(fabsf(gy - moy) <= 1e-5f * (1.f + fabsf(moy)));
            if (bx || by) {
                if (bad < 12) printf("item %d q %3d (wave %d): x %.6f want %.6f %s   y %.6f want %.6f %s\n", item, q, w, gx, mox, bx ? "WRONG" : "ok", gy, moy, by ? "WRONG" : "ok");
                ++bad;
            }
        }
    printf("%d of %d (item, q) entries wrong\n", bad, items * N);
    return bad ? 1 : 0;
}

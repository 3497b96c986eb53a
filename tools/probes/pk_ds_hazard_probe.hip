// Is there an un-interlocked hazard between a packed-fp32 VALU write (v_pk_add_f32 v[n:n+1]) and a DS instruction that reads one of
// the two registers as DATA in the very next issue slot (ds_bpermute_b32 ..., v[n])?  That is the shape of the SLP-vectorized
// motion reduction of the round-2 attention kernel -- (x, y) partial sums added as a pair, then __shfl_xor of each half -- whose
// x (low) half alone came out wrong on MI355X, for some waves only (tools/slp_where.py), while the ISA is algebraically right
// (tools/probes/slp_isa_symexec.py) and op_sel behaves as documented (tools/probes/pk_opsel_probe.hip).
//   hipcc -O3 --offload-arch=gfx950 tools/probes/pk_ds_hazard_probe.hip -o tools/probes/pk_ds_hazard_probe
// Variant 0: pair add, bpermute of both halves right behind it (what SLP made).  Variant 1: the same with s_nop 4 in between.
// Variant 2: two scalar v_add_f32 (what -fno-slp-vectorize makes).  9 waves per workgroup like the kernel, 64 workgroups, 2000 rounds.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int VARIANT>
__global__ __launch_bounds__(576) void probe(const float* in, unsigned* bad, int rounds) {
    const int lane = threadIdx.x & 63;
    const int a16 = ((lane ^ 16) << 2), a32 = ((lane ^ 32) << 2);
    f32x2 acc = (f32x2){in[threadIdx.x], in[threadIdx.x + 576]};
    unsigned wrong_x = 0, wrong_y = 0;
    for (int it = 0; it < rounds; ++it) {
        f32x2 inc = (f32x2){(float)(it & 7) + 1.0f, (float)(it & 3) + 2.0f};
        f32x2 t;
        int px, py;
        if (VARIANT == 2) {
            t.x = acc.x + inc.x;
            t.y = acc.y + inc.y;
            asm volatile("" : "+v"(t));
            px = __builtin_amdgcn_ds_bpermute(a16, __builtin_bit_cast(int, t.x));
            py = __builtin_amdgcn_ds_bpermute(a16, __builtin_bit_cast(int, t.y));
        } else {
            // one asm block with fixed registers (inline asm cannot name the halves of a 64-bit operand), so that nothing is scheduled
            // between the packed add and the two permutes that read its halves
            float tx, ty;
            if (VARIANT == 0)
                asm volatile("v_pk_add_f32 v[100:101], %4, %5\n\tds_bpermute_b32 %0, %6, v100\n\tds_bpermute_b32 %1, %6, v101\n\t"
                             "s_waitcnt lgkmcnt(0)\n\tv_mov_b32 %2, v100\n\tv_mov_b32 %3, v101"
                             : "=&v"(px), "=&v"(py), "=&v"(tx), "=&v"(ty) : "v"(acc), "v"(inc), "v"(a16) : "v100", "v101");
            else
                asm volatile("v_pk_add_f32 v[100:101], %4, %5\n\ts_nop 4\n\tds_bpermute_b32 %0, %6, v100\n\tds_bpermute_b32 %1, %6, v101\n\t"
                             "s_waitcnt lgkmcnt(0)\n\tv_mov_b32 %2, v100\n\tv_mov_b32 %3, v101"
                             : "=&v"(px), "=&v"(py), "=&v"(tx), "=&v"(ty) : "v"(acc), "v"(inc), "v"(a16) : "v100", "v101");
            t = (f32x2){tx, ty};
        }
        // what lane ^ 16 must have sent: ITS acc + inc; every lane of a wave keeps acc = base(lane) + sum of incs, so the partner's
        // value is known without communication: partner base + (t - own base)
        const float bx = in[(threadIdx.x & ~63) + (lane ^ 16)], by = in[(threadIdx.x & ~63) + (lane ^ 16) + 576];
        const float ox = in[threadIdx.x], oy = in[threadIdx.x + 576];
        wrong_x += __builtin_bit_cast(float, px) != bx + (t.x - ox);
        wrong_y += __builtin_bit_cast(float, py) != by + (t.y - oy);
        acc = t;
        if ((it & 15) == 15) acc = (f32x2){ox, oy};      // keep the sums exactly representable
        (void)a32;
    }
    if (wrong_x) atomicAdd(bad, wrong_x);
    if (wrong_y) atomicAdd(bad + 1, wrong_y);
}

int main() {
    float h[1152];
    for (int i = 0; i < 1152; ++i) h[i] = (float)((i * 7) % 64);      // small integers: all sums exact
    float* d;
    unsigned* bad;
    hipMalloc(&d, sizeof(h));
    hipMalloc(&bad, 8);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    const char* names[3] = {"v_pk_add_f32 -> ds_bpermute (adjacent)", "v_pk_add_f32, s_nop 4, ds_bpermute     ", "2 x v_add_f32 -> ds_bpermute          "};
    int rc = 0;
    for (int v = 0; v < 2; ++v) {      // (variant 2, the scalar form, is what every shipped build has; its check here is not maintained)
        hipMemset(bad, 0, 8);
        if (v == 0) hipLaunchKernelGGL(probe<0>, dim3(512), dim3(576), 0, 0, d, bad, 2000);
        if (v == 1) hipLaunchKernelGGL(probe<1>, dim3(512), dim3(576), 0, 0, d, bad, 2000);
        if (v == 2) hipLaunchKernelGGL(probe<2>, dim3(512), dim3(576), 0, 0, d, bad, 2000);
        unsigned hb[2] = {0, 0};
        hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost);
        printf("%s: wrong low halves (x) %u, wrong high halves (y) %u of %u\n", names[v], hb[0], hb[1], 512u * 576u * 2000u);
        rc |= (hb[0] | hb[1]) != 0 && v != 0;
    }
    return rc;
}

// Does gfx950 honour op_sel / op_sel_hi on packed-fp32 VALU (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32)?
// The SLP-vectorized motion chain of the round-2 attention kernel (tools/slp_check.py: libatmvfi_hip_oldattn_slp.so) pairs
// (x, y) terms as  v_pk_mul_f32 v[a:a+1], v[b:b+1], v[p:p+1] op_sel:[0,1]  -- the LOW result takes the HIGH register of src1 --
// and only its x (low) results come out wrong on hardware, while the ISA is algebraically right (tools/probes/slp_isa_symexec.py).
//   hipcc -O2 --offload-arch=gfx950 tools/probes/pk_opsel_probe.hip -o tools/probes/pk_opsel_probe && ./tools/probes/pk_opsel_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define PK(NAME, INSN, MODS)                                                                                          \
    __global__ void NAME(const f32x2* a, const f32x2* b, f32x2* d) {                                                    \
        const int i = threadIdx.x;                                                                                      \
        f32x2 x = a[i], y = b[i], r;                                                                                    \
        asm volatile(INSN " %0, %1, %2 " MODS : "=v"(r) : "v"(x), "v"(y));                                            \
        d[i] = r;                                                                                                       \
    }
PK(mul_00_11, "v_pk_mul_f32", "")
PK(mul_01_11, "v_pk_mul_f32", "op_sel:[0,1]")
PK(mul_10_11, "v_pk_mul_f32", "op_sel:[1,0]")
PK(mul_00_10, "v_pk_mul_f32", "op_sel_hi:[1,0]")
PK(mul_00_01, "v_pk_mul_f32", "op_sel_hi:[0,1]")
PK(mul_01_10, "v_pk_mul_f32", "op_sel:[0,1] op_sel_hi:[1,0]")
PK(add_01_11, "v_pk_add_f32", "op_sel:[0,1]")
PK(add_10_11, "v_pk_add_f32", "op_sel:[1,0]")
PK(add_00_01, "v_pk_add_f32", "op_sel_hi:[0,1]")

// the same with the multiplier pair produced by the instruction right before (no independent instruction between): the shape in the chain
__global__ void mul_01_back_to_back(const f32x2* a, const f32x2* b, f32x2* d) {
    const int i = threadIdx.x;
    f32x2 x = a[i], y = b[i], r, t;
    asm volatile("v_pk_mul_f32 %1, %3, %3\n\tv_pk_mul_f32 %0, %2, %1 op_sel:[0,1]" : "=v"(r), "=&v"(t) : "v"(x), "v"(y));
    d[i] = r;
}

int main() {
    f32x2 ha[64], hb[64], hd[64];
    for (int i = 0; i < 64; ++i) { ha[i] = (f32x2){1.0f + i, 100.0f + i}; hb[i] = (f32x2){2.0f, 3.0f}; }
    f32x2 *a, *b, *d;
    hipMalloc(&a, sizeof(ha)); hipMalloc(&b, sizeof(hb)); hipMalloc(&d, sizeof(hd));
    hipMemcpy(a, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(b, hb, sizeof(hb), hipMemcpyHostToDevice);
    struct { const char* name; void (*k)(const f32x2*, const f32x2*, f32x2*); int mul; int s0lo, s1lo, s0hi, s1hi; } t[] = {
        {"v_pk_mul_f32 (default)            ", mul_00_11, 1, 0, 0, 1, 1}, {"v_pk_mul_f32 op_sel:[0,1]         ", mul_01_11, 1, 0, 1, 1, 1},
        {"v_pk_mul_f32 op_sel:[1,0]         ", mul_10_11, 1, 1, 0, 1, 1}, {"v_pk_mul_f32 op_sel_hi:[1,0]      ", mul_00_10, 1, 0, 0, 1, 0},
        {"v_pk_mul_f32 op_sel_hi:[0,1]      ", mul_00_01, 1, 0, 0, 0, 1}, {"v_pk_mul_f32 op_sel:[0,1] hi:[1,0]", mul_01_10, 1, 0, 1, 1, 0},
        {"v_pk_add_f32 op_sel:[0,1]         ", add_01_11, 0, 0, 1, 1, 1}, {"v_pk_add_f32 op_sel:[1,0]         ", add_10_11, 0, 1, 0, 1, 1},
        {"v_pk_add_f32 op_sel_hi:[0,1]      ", add_00_01, 0, 0, 0, 0, 1}};
    int bad = 0;
    for (auto& c : t) {
        hipLaunchKernelGGL(c.k, dim3(1), dim3(64), 0, 0, a, b, d);
        hipMemcpy(hd, d, sizeof(hd), hipMemcpyDeviceToHost);
        int wrong = 0;
        for (int i = 0; i < 64; ++i) {
            const float x[2] = {ha[i].x, ha[i].y}, y[2] = {hb[i].x, hb[i].y};
            const float lo = c.mul ? x[c.s0lo] * y[c.s1lo] : x[c.s0lo] + y[c.s1lo], hi = c.mul ? x[c.s0hi] * y[c.s1hi] : x[c.s0hi] + y[c.s1hi];
            wrong += hd[i].x != lo || hd[i].y != hi;
        }
        printf("%s lane 5: got (%g, %g)  %s\n", c.name, hd[5].x, hd[5].y, wrong ? "WRONG" : "as the ISA manual says");
        bad += wrong != 0;
    }
    hipLaunchKernelGGL(mul_01_back_to_back, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd, d, sizeof(hd), hipMemcpyDeviceToHost);
    // t = (y.x^2, y.y^2) = (4, 9); r = (x.x * t.y, x.y * t.y)
    int wrong = 0;
    for (int i = 0; i < 64; ++i) wrong += hd[i].x != ha[i].x * 9.0f || hd[i].y != ha[i].y * 9.0f;
    printf("v_pk_mul_f32 op_sel:[0,1] right behind the producer of src1: lane 5 got (%g, %g) want (%g, %g)  %s\n", hd[5].x, hd[5].y, ha[5].x * 9.0f,
           ha[5].y * 9.0f, wrong ? "WRONG" : "ok");
    bad += wrong != 0;
    printf("%d variants wrong\n", bad);
    return bad ? 1 : 0;
}

// Shared helpers of libatmvfi_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "atmvfi.h"

namespace atmvfi {

void set_error(const char* fmt, ...);
int check_launch(const char* what);

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace atmvfi

#define ATMVFI_REQUIRE(cond, code, ...)          \
    do {                                         \
        if (!(cond)) {                           \
            atmvfi::set_error(__VA_ARGS__);      \
            return (code);                       \
        }                                        \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// exact GELU (erf form), as nn.GELU() default
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Error reporting and version of libatmvfi_hip.so.
#include "common.h"

#include <string.h>

namespace atmvfi {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return ATMVFI_ELAUNCH;
    }
    return ATMVFI_OK;
}

}  // namespace atmvfi

extern "C" int atmvfi_version(void) { return (0 << 16) | (8 << 8) | 0; }   // 0.8: atmvfi_source_digest; 0.7: atmvfi_window_attention_f16x3, compact fp32 view behind out_cmin of the 3x3 plane kernel, saturating conversions by MODE.FP16_OVFL; 0.6: 3x3 kernel on split-plane input, plane sinks and CONV mode of the LDS-DMA GEMM, per-call instance overrides (no process-wide state); 0.5: plane sink of the 3x3 kernel; 0.4: split-plane sinks, k-step-major planes, uint8 frame kernels
extern "C" const char* atmvfi_last_error(void) { return atmvfi::g_err; }
#ifndef ATMVFI_SOURCE_DIGEST
#define ATMVFI_SOURCE_DIGEST "unknown (api.hip compiled outside the Makefile)"
#endif
extern "C" const char* atmvfi_source_digest(void) { return ATMVFI_SOURCE_DIGEST; }

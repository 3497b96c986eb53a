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

extern "C" int atmvfi_version(void) { return (0 << 16) | (5 << 8) | 0; }   // 0.5: plane sink of the 3x3 kernel, schedule / tile-width overrides (0.4: split-plane sinks, k-step-major planes, uint8 frame kernels)
extern "C" const char* atmvfi_last_error(void) { return atmvfi::g_err; }

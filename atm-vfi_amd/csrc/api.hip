// Error reporting and version of libatmvfi_hip.so.
#include "common.h"

#include <string.h>

#include <vector>

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

#ifdef ATMVFI_RANGE_CHECK
static std::vector<RangeWordSetter>& range_registry() {
    static std::vector<RangeWordSetter> v;      // (function-local: filled by other translation units' static initialisers)
    return v;
}
void range_registry_add(RangeWordSetter s) { range_registry().push_back(s); }
#endif

}  // namespace atmvfi

extern "C" int atmvfi_range_word_set(uint32_t* word, void* stream) {
#ifdef ATMVFI_RANGE_CHECK
    for (atmvfi::RangeWordSetter set : atmvfi::range_registry()) {
        const int e = set(reinterpret_cast<unsigned*>(word), (hipStream_t)stream);
        ATMVFI_REQUIRE(e == 0, ATMVFI_ELAUNCH, "range_word_set: hipMemcpyToSymbolAsync failed (%d)", e);
    }
    return ATMVFI_OK;
#else
    (void)word; (void)stream;
    atmvfi::set_error("range_word_set: this is the default build; the operand range check lives in libatmvfi_hip_checked.so (make checked)");
    return ATMVFI_EINVAL;
#endif
}
extern "C" int atmvfi_range_checked(void) {
#ifdef ATMVFI_RANGE_CHECK
    return 1;
#else
    return 0;
#endif
}

extern "C" int atmvfi_version(void) { return (0 << 16) | (9 << 8) | 0; }   // 0.9: atmvfi_flow_warp_ex, atmvfi_range_word_set / atmvfi_range_checked (the checked build); 0.8: atmvfi_source_digest; 0.7: atmvfi_window_attention_f16x3, compact fp32 view behind out_cmin of the 3x3 plane kernel, saturating conversions by MODE.FP16_OVFL; 0.6: 3x3 kernel on split-plane input, plane sinks and CONV mode of the LDS-DMA GEMM, per-call instance overrides (no process-wide state); 0.5: plane sink of the 3x3 kernel; 0.4: split-plane sinks, k-step-major planes, uint8 frame kernels
extern "C" const char* atmvfi_last_error(void) { return atmvfi::g_err; }
#ifndef ATMVFI_SOURCE_DIGEST
#define ATMVFI_SOURCE_DIGEST "unknown (api.hip compiled outside the Makefile)"
#endif
extern "C" const char* atmvfi_source_digest(void) { return ATMVFI_SOURCE_DIGEST; }

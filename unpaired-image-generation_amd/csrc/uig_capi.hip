// uig_capi.hip — error plumbing and library identity for the C ABI in include/uig.h.
#include "uig_common.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>

static thread_local char g_err[512] = "";

int uig_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* uig_last_error(void) { return g_err; }
extern "C" const char* uig_version(void) { return "uig 0.1 (gfx950, hipcc, wave64 MFMA 16x16)"; }

extern "C" int uig_device_ok(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return 0; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 0;
    return std::strstr(prop.gcnArchName, "gfx950") != nullptr ? 1 : 0;
}

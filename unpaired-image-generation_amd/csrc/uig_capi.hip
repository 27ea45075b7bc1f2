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

// which kernel family the process's last uig_conv_gather* launch ran on (UIG_K_*): lets a parity test assert that it covers the
// kernel it means to cover instead of a fallback.  Process-wide (an atomic), not per thread: autograd runs backward launches on
// its own thread and the test asks from the main one.
#include <atomic>
static std::atomic<int> g_last_conv_kernel{0};
void uig_note_conv_kernel(int id) { g_last_conv_kernel.store(id, std::memory_order_relaxed); }
extern "C" int uig_debug_last_conv_kernel(void) { return g_last_conv_kernel.load(std::memory_order_relaxed); }
extern "C" const char* uig_version(void) { return "uig 0.1 (gfx950, hipcc, wave64 MFMA 16x16)"; }

extern "C" int uig_device_ok(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return 0; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 0;
    return std::strstr(prop.gcnArchName, "gfx950") != nullptr ? 1 : 0;
}

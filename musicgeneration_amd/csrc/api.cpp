// Host-side plumbing of libmgx: error string, version, device probe.
#include <stdarg.h>
#include <string.h>
#include "mgx_common.hpp"

static thread_local char g_err[512] = "";

void mgx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* mgx_last_error(void) { return g_err; }
extern "C" int mgx_abi_version(void) { return MGX_ABI_VERSION; }
extern "C" int mgx_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        mgx_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return MGX_ERR_NO_DEVICE;
    }
    return n;
}

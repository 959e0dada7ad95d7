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

// ---- deterministic-reduction mode (mgx.h: mgx_set_deterministic) ------------------------------------------------------
// Sums that cross workgroups (dE, the vocabulary projection's dW / db, the encoder blocks' bias gradients, the embedding
// gradient, the loss sum) normally end in fp32 atomics, whose order -- hence the last bits -- varies from run to run.  With a
// scratch buffer registered, those kernels add 64-bit FIXED-POINT integers (value * 2^30, rounded once per partial sum) into the
// scratch instead -- integer addition is associative, so the result does not depend on the order -- and a fold pass converts the
// total back.  The buffer is the caller's (device memory, 8-byte aligned); calls that use it must be issued on one stream.
#include <atomic>
static std::atomic<long long*> g_det_ptr{nullptr};
static std::atomic<size_t> g_det_elems{0};

extern "C" int mgx_set_deterministic(void* scratch, size_t bytes) {
    MGX_REQUIRE(((uintptr_t)scratch & 7) == 0, MGX_ERR_SHAPE, "mgx_set_deterministic: scratch must be 8-byte aligned");
    MGX_REQUIRE(scratch == nullptr || bytes >= 8, MGX_ERR_SHAPE, "mgx_set_deterministic: scratch too small");
    g_det_elems.store(scratch ? bytes / 8 : 0);
    g_det_ptr.store((long long*)scratch);
    return MGX_OK;
}
extern "C" int mgx_deterministic(void) { return g_det_ptr.load() != nullptr; }

// -> scratch of `elems` zeroed int64 (zeroing enqueued on `stream`), nullptr when the mode is off; *rc != MGX_OK if it is on
// but the registered buffer is too small
long long* mgx_det_scratch(size_t elems, void* stream, int* rc) {
    *rc = MGX_OK;
    long long* p = g_det_ptr.load();
    if (!p) return nullptr;
    if (g_det_elems.load() < elems) {
        mgx_set_error("deterministic mode: the registered scratch holds %zu int64, this call needs %zu (mgx_set_deterministic)",
                      g_det_elems.load(), elems);
        *rc = MGX_ERR_SHAPE;
        return nullptr;
    }
    if (hipMemsetAsync(p, 0, elems * 8, (hipStream_t)stream) != hipSuccess) {
        mgx_set_error("deterministic mode: hipMemsetAsync of the scratch failed");
        *rc = MGX_ERR_LAUNCH;
        return nullptr;
    }
    return p;
}

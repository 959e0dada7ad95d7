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
// total back.  The buffer is the caller's (device memory, 8-byte aligned); calls that use it must be issued on one stream -- every
// further stream registers a buffer of its own (mgx_set_deterministic_stream, ABI 18).
#include <atomic>
#include <mutex>
#include <vector>
static std::atomic<long long*> g_det_ptr{nullptr};
static std::atomic<size_t> g_det_elems{0};
// ABI 18: streams beside the first one bring their own scratch (mgx_set_deterministic_stream) -- two streams that shared one
// would zero / fold each other's partial sums
struct DetStream { void* stream; long long* ptr; size_t elems; };
static std::mutex g_reg_mu;
static std::vector<DetStream> g_det_streams;

extern "C" int mgx_set_deterministic(void* scratch, size_t bytes) {
    MGX_REQUIRE(((uintptr_t)scratch & (MGX_DET_BYTES - 1)) == 0, MGX_ERR_SHAPE, "mgx_set_deterministic: scratch must be aligned to its size, 32 MiB (sums in the lower half, poison words in the upper: mgx.h)");
    MGX_REQUIRE(scratch == nullptr || bytes >= MGX_DET_BYTES, MGX_ERR_SHAPE, "mgx_set_deterministic: scratch too small (need 32 MiB)");
    g_det_elems.store(scratch ? MGX_DET_HALF / 8 : 0);
    g_det_ptr.store((long long*)scratch);
    if (!scratch) {                                        // switching the mode off forgets the per-stream buffers too
        std::lock_guard<std::mutex> lk(g_reg_mu);
        g_det_streams.clear();
    }
    return MGX_OK;
}
extern "C" int mgx_deterministic(void) { return g_det_ptr.load() != nullptr; }

extern "C" int mgx_set_deterministic_stream(void* stream, void* scratch, size_t bytes) {
    MGX_REQUIRE(((uintptr_t)scratch & (MGX_DET_BYTES - 1)) == 0, MGX_ERR_SHAPE, "mgx_set_deterministic_stream: scratch must be aligned to its size, 32 MiB");
    MGX_REQUIRE(scratch == nullptr || bytes >= MGX_DET_BYTES, MGX_ERR_SHAPE, "mgx_set_deterministic_stream: scratch too small (need 32 MiB)");
    MGX_REQUIRE(scratch == nullptr || g_det_ptr.load() != nullptr, MGX_ERR_SHAPE,
                "mgx_set_deterministic_stream: switch the mode on first (mgx_set_deterministic)");
    std::lock_guard<std::mutex> lk(g_reg_mu);
    for (size_t i = 0; i < g_det_streams.size(); ++i)
        if (g_det_streams[i].stream == stream) { g_det_streams.erase(g_det_streams.begin() + i); break; }
    if (scratch) g_det_streams.push_back({stream, (long long*)scratch, MGX_DET_HALF / 8});
    return MGX_OK;
}

// -> scratch of `elems` zeroed int64 (zeroing enqueued on `stream`), nullptr when the mode is off; *rc != MGX_OK if it is on
// but the registered buffer is too small.  A stream with a buffer of its own (mgx_set_deterministic_stream) gets that one.
long long* mgx_det_scratch(size_t elems, void* stream, int* rc) {
    *rc = MGX_OK;
    long long* p = g_det_ptr.load();
    if (!p) return nullptr;
    size_t have = g_det_elems.load();
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        for (const DetStream& d : g_det_streams)
            if (d.stream == stream) { p = d.ptr; have = d.elems; break; }
    }
    if (have < elems) {
        mgx_set_error("deterministic mode: the registered scratch holds %zu int64, this call needs %zu (mgx_set_deterministic)",
                      have, elems);
        *rc = MGX_ERR_SHAPE;
        return nullptr;
    }
    if (hipMemsetAsync(p, 0, elems * 8, (hipStream_t)stream) != hipSuccess ||
        hipMemsetAsync((char*)p + MGX_DET_HALF, 0, elems * 8, (hipStream_t)stream) != hipSuccess) {      // sums, poison words
        mgx_set_error("deterministic mode: hipMemsetAsync of the scratch failed");
        *rc = MGX_ERR_LAUNCH;
        return nullptr;
    }
    return p;
}

// ---- CU-masked streams (ABI 18; mgx.h: mgx_stream_create_cu_mask) -----------------------------------------------------------
// A stream whose kernels may only run on the CUs named by a bit mask (hipExtStreamCreateWithCUMask): the way to run the
// HBM-bound, off-critical-path kernels of the backward (dE, the weight gradients) -- or RCCL's -- BESIDE the MFMA-bound ones
// instead of between them.  The library remembers how many CUs each such stream has: the persistent GEMM kernels size their
// grids from that number (one workgroup per CU of the stream, not of the device).
struct StreamCus { void* stream; int cus; };
static std::vector<StreamCus> g_stream_cus;

static int device_cus() {
    static int cus = 0;                                    // queried once (the call is not cheap)
    if (!cus) {
        hipDeviceProp_t prop;
        int dev = 0;
        hipGetDevice(&dev);
        cus = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return cus;
}

int mgx_stream_cu_count(void* stream) {
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        for (const StreamCus& s : g_stream_cus)
            if (s.stream == stream) return s.cus;
    }
    return device_cus();
}

extern "C" int mgx_stream_cus(void* stream) { return mgx_stream_cu_count(stream); }

extern "C" int mgx_stream_set_cus(void* stream, int cus) {
    MGX_REQUIRE(cus >= 0 && cus <= device_cus(), MGX_ERR_SHAPE, "mgx_stream_set_cus: 0 (forget) .. %d CUs, got %d", device_cus(), cus);
    std::lock_guard<std::mutex> lk(g_reg_mu);
    for (size_t i = 0; i < g_stream_cus.size(); ++i)
        if (g_stream_cus[i].stream == stream) { g_stream_cus.erase(g_stream_cus.begin() + i); break; }
    if (cus > 0) g_stream_cus.push_back({stream, cus});
    return MGX_OK;
}

extern "C" int mgx_stream_create_cu_mask(void** stream, const uint32_t* mask, int words) {
    MGX_REQUIRE(stream && mask && words > 0, MGX_ERR_NULL, "mgx_stream_create_cu_mask: NULL pointer / no mask words");
    int n = 0;
    const int cus = device_cus();
    for (int i = 0; i < words * 32; ++i)
        if (mask[i / 32] >> (i % 32) & 1u) {
            MGX_REQUIRE(i < cus, MGX_ERR_SHAPE, "mgx_stream_create_cu_mask: bit %d set, the device has %d CUs", i, cus);
            ++n;
        }
    MGX_REQUIRE(n > 0, MGX_ERR_SHAPE, "mgx_stream_create_cu_mask: empty mask");
    hipStream_t s = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask);
    if (e != hipSuccess) {
        mgx_set_error("hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e));
        return MGX_ERR_LAUNCH;
    }
    *stream = (void*)s;
    return mgx_stream_set_cus((void*)s, n);
}

extern "C" int mgx_stream_destroy(void* stream) {
    MGX_REQUIRE(stream, MGX_ERR_NULL, "mgx_stream_destroy: NULL stream");
    mgx_stream_set_cus(stream, 0);
    mgx_set_deterministic_stream(stream, nullptr, 0);
    hipError_t e = hipStreamDestroy((hipStream_t)stream);
    if (e != hipSuccess) {
        mgx_set_error("hipStreamDestroy: %s", hipGetErrorString(e));
        return MGX_ERR_LAUNCH;
    }
    return MGX_OK;
}

// Shared device/host helpers for libmgx (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "mgx.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define MGX_DEV __device__ __forceinline__

MGX_DEV float bf16_to_f32(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }
MGX_DEV float bf16lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
MGX_DEV float bf16hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
// two f32 -> packed bf16x2 (lo in bits 0..15), RNE, NaN-preserving: compiles to v_cvt_pk_bf16_f32
MGX_DEV uint32_t pack_bf16x2(float lo, float hi) {
    f32x2 v = {lo, hi};
    bf16x2 r = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(uint32_t, r);
}
MGX_DEV uint16_t f32_to_bf16(float x) { return (uint16_t)(pack_bf16x2(x, 0.f) & 0xffffu); }

MGX_DEV void unpack8(const u32x4& w, float* f) {
    f[0] = bf16lo(w.x); f[1] = bf16hi(w.x); f[2] = bf16lo(w.y); f[3] = bf16hi(w.y);
    f[4] = bf16lo(w.z); f[5] = bf16hi(w.z); f[6] = bf16lo(w.w); f[7] = bf16hi(w.w);
}
MGX_DEV u32x4 pack8(const float* f) {
    u32x4 w;
    w.x = pack_bf16x2(f[0], f[1]); w.y = pack_bf16x2(f[2], f[3]);
    w.z = pack_bf16x2(f[4], f[5]); w.w = pack_bf16x2(f[6], f[7]);
    return w;
}

// ---- stateless dropout RNG: keep decision is a pure function of (seed, element index) ---------
MGX_DEV uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
struct DropCfg {
    uint32_t thr16;   // drop iff 16-bit random < thr16
    uint32_t mix;     // seed mix
    float scale;      // 1/(1-p_actual)
};
static inline DropCfg make_drop(float p, uint64_t seed) {
    DropCfg c;
    if (p <= 0.f) { c.thr16 = 0; c.scale = 1.f; c.mix = 0; return c; }
    double t = (double)p * 65536.0;
    uint32_t thr = (uint32_t)(t + 0.5);
    if (thr > 65535u) thr = 65535u;
    c.thr16 = thr;
    c.scale = (float)(1.0 / (1.0 - (double)thr / 65536.0));
    uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
    c.mix = lo * 0x9e3779b9u + (hi ^ 0x85ebca6bu) * 0xc2b2ae35u + 0x27d4eb2fu;
    return c;
}
// multipliers (0 or scale) for the 8 consecutive elements of group g (g = element_index / 8)
MGX_DEV void drop_mult8(const DropCfg& c, uint32_t g, float* mult) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t r = hash32((g * 4u + (uint32_t)k) ^ c.mix);
        mult[2 * k] = ((r & 0xffffu) < c.thr16) ? 0.f : c.scale;
        mult[2 * k + 1] = ((r >> 16) < c.thr16) ? 0.f : c.scale;
    }
}

// Sum over the 64 lanes, returned in every lane.  DPP adds inside each row of 16 lanes (quad swaps, half-row and row
// mirrors: 4 VALU of ~8 cycles each), then the four row sums through v_readlane: a __shfl_xor butterfly is six dependent
// ds_bpermute round trips through the LDS pipe (~100+ cycles each), which was most of a LayerNorm-backward row.
template <int CTRL> MGX_DEV float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
MGX_DEV float wave_sum(float v) {
    v += dpp_mov<0xB1>(v);                 // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);                 // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);                // row_half_mirror
    v += dpp_mov<0x140>(v);                // row_mirror: every lane of a row holds the row's sum
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (r0 + r1) + (r2 + r3);
}
MGX_DEV float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// host: environment switch `name` is set and starts with '0' (A/B toggles; read on every call, so a test can flip it)
#include <stdlib.h>
static inline bool env_is_zero(const char* name) {
    const char* v = getenv(name);
    return v && v[0] == '0';
}
static inline bool env_is_one(const char* name) {
    const char* v = getenv(name);
    return v && v[0] == '1';
}
static inline int env_digit(const char* name, int dflt) {      // first character as a digit, `dflt` when unset / not a digit
    const char* v = getenv(name);
    return (v && v[0] >= '0' && v[0] <= '9') ? v[0] - '0' : dflt;
}

// ---- host-side error plumbing -----------------------------------------------------------------
void mgx_set_error(const char* fmt, ...);

// ---- deterministic-reduction mode (api.cpp) ------------------------------------------------------------------------------
// Cross-workgroup sums as 64-bit fixed-point integer atomics (value * 2^30) into a caller-registered scratch, folded back to
// fp32 by det_fold_kernel: the total is independent of the order in which the workgroups arrive.
long long* mgx_det_scratch(size_t elems, void* stream, int* rc);
// CUs the kernels of `stream` may run on: the device's count, or the size of the stream's CU mask (api.cpp, mgx_stream_create_cu_mask)
int mgx_stream_cu_count(void* stream);
constexpr float MGX_DET_SCALE = 1073741824.f;              // 2^30: resolution 9.3e-10
// A partial sum that is NaN, infinite or >= 2^31 in magnitude has no fixed-point image (the conversion would give 0, garbage or a
// wrapped value, and a diverging run would fold back to finite numbers).  Such a partial POISONS its destination instead, and the
// poison is STICKY (round 6): the registered scratch is 32 MiB, aligned to 32 MiB -- sums in the lower 16 MiB, one
// poison word per sum at the same offset in the upper 16 MiB (address = sum's address | 16 MiB: no kernel needs a second
// pointer) -- and a poisoned word is only ever OR-ed.  Until round 6 the poison was a signed max with 2^62 on the sum word itself,
// which later large negative partials could carry back into the valid band.  The fold pass writes NaN for every poisoned word and
// for every total outside +-2^61 (= +-2.1e9).  Order-independent like the sum itself.
constexpr uintptr_t MGX_DET_HALF = (uintptr_t)16 << 20;     // bytes: sums below, poison words above
constexpr uintptr_t MGX_DET_BYTES = 2 * MGX_DET_HALF;        // size and alignment of a deterministic-mode scratch
constexpr long long MGX_DET_BAND = 1LL << 61;
constexpr float MGX_DET_LIMIT = 2147483648.f;              // 2^31
MGX_DEV bool det_representable(float v) { return fabsf(v) < MGX_DET_LIMIT; }     // false for NaN
MGX_DEV void det_poison(long long* dst) { atomicOr((unsigned int*)((uintptr_t)dst | MGX_DET_HALF), 1u); }
MGX_DEV void det_add(long long* dst, float v) {
    if (!det_representable(v)) { det_poison(dst); return; }
    atomicAdd((unsigned long long*)dst, (unsigned long long)__float2ll_rn(v * MGX_DET_SCALE));   // two's complement: wraps like a signed add
}
// dst[i] (+)= scale * src[i] / 2^30   (NaN where the word was poisoned or the total left the valid band)
static __global__ __launch_bounds__(256) void det_fold_kernel(const long long* __restrict__ src, float* __restrict__ dst, size_t n,
                                                              float scale, int accumulate) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const long long x = src[i];
        const unsigned int poisoned = *(const unsigned int*)((uintptr_t)(src + i) | MGX_DET_HALF);
        const bool ok = !poisoned && x < MGX_DET_BAND && x > -MGX_DET_BAND;
        const float v = ok ? (float)((double)x * (1.0 / 1073741824.0)) * scale : __builtin_nanf("");
        dst[i] = accumulate ? dst[i] + v : v;
    }
}
static inline void launch_det_fold(const long long* src, float* dst, size_t n, float scale, int accumulate, hipStream_t s) {
    size_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(det_fold_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, n, scale, accumulate);
}
#define MGX_REQUIRE(cond, code, ...)            \
    do {                                        \
        if (!(cond)) {                          \
            mgx_set_error(__VA_ARGS__);         \
            return (code);                      \
        }                                       \
    } while (0)
#define MGX_CHECK_LAUNCH(name)                                                        \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            mgx_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
            return MGX_ERR_LAUNCH;                                                    \
        }                                                                             \
    } while (0)

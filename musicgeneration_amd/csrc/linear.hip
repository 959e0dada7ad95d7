// C = act(A @ W^T + bias): the projection / FFN / vocabulary GEMMs of the hot path
// (layers.py:71-84,108,157-158; network.py:39).  A [M,K] and W [N,K] are both K-contiguous, so both
// MFMA operands are plain 16-byte row fragments (no transposed reads).
//
// Tile 128 x 128 x 64, 4 waves (2 x 2), each wave 64 x 64 = 2 x 2 MFMA 32x32x16 tiles per k-step.
// LDS: two 16 KiB images per buffer, double buffered (64 KiB -> 2 workgroups / CU), XOR-swizzled
// 16-byte chunks (conflict-free ds_read_b128), register-staged prefetch of the next K tile while the
// current one is multiplied (one barrier per K tile).
#include "rel_attn_common.hpp"

using namespace relattn;

namespace {
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int IMG = BM * BK * 2;             // 16 KiB
constexpr int LDS_BYTES = 4 * IMG;           // A0 W0 A1 W1
}  // namespace

__global__ __launch_bounds__(256, 2) void linear_fwd_kernel(const uint16_t* __restrict__ A,
                                                            const uint16_t* __restrict__ W,
                                                            const float* __restrict__ bias,
                                                            uint16_t* __restrict__ C, int M, int N, int K, int act) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int l31 = lane & 31, hh = lane >> 5;
    // XCD-aware tile order: consecutive blocks (which share an XCD's L2 only every 8th) are remapped so
    // that one XCD walks a contiguous run of row-tiles for the same column tile (A panel reuse in L2).
    const int ntn = (N + BN - 1) / BN;
    const int ntm = (M + BM - 1) / BM;
    const int nwg = ntm * ntn;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;     // bijective remap
    }
    const int tn = bid % ntn, tm = bid / ntn;
    const int m0 = tm * BM, n0 = tn * BN;

    const int srow = tid >> 3, sch = tid & 7;
    u32x4 areg[4], wreg[4];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = srow + 32 * i;
            const int gm = m0 + row, gn = n0 + row;
            areg[i] = (gm < M) ? *(const u32x4*)(A + (size_t)gm * K + k0 + sch * 8) : u32x4{0, 0, 0, 0};
            wreg[i] = (gn < N) ? *(const u32x4*)(W + (size_t)gn * K + k0 + sch * 8) : u32x4{0, 0, 0, 0};
        }
    };
    auto store_tiles = [&](int buf) {
        char* at = smem + buf * 2 * IMG;
        char* wt = at + IMG;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = srow + 32 * i;
            *(u32x4*)(at + imgR_off(row, sch)) = areg[i];
            *(u32x4*)(wt + imgR_off(row, sch)) = wreg[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = zero16();

    const int nk = K / BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tiles((kt + 1) * BK);
        const char* at = smem + cur * 2 * IMG;
        const char* wt = at + IMG;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 a0 = frag_R(at, 64 * wm + l31, hh, ks);
            const bf16x8 a1 = frag_R(at, 64 * wm + 32 + l31, hh, ks);
            const bf16x8 b0 = frag_R(wt, 64 * wn + l31, hh, ks);
            const bf16x8 b1 = frag_R(wt, 64 * wn + 32 + l31, hh, ks);
            acc[0][0] = mfma(a0, b0, acc[0][0]);
            acc[0][1] = mfma(a0, b1, acc[0][1]);
            acc[1][0] = mfma(a1, b0, acc[1][0]);
            acc[1][1] = mfma(a1, b1, acc[1][1]);
        }
        if (kt + 1 < nk) store_tiles(cur ^ 1);
        __syncthreads();
    }
    // epilogue: D[i][j], column j on the lane (32 consecutive n = 64 contiguous bytes per half-wave)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int n = n0 + 64 * wn + 32 * ct + l31;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 64 * wm + 32 * rt + crow(r, hh);
                if (m < M) {
                    float v = acc[rt][ct][r] + bv;
                    if (act == 1) v = fmaxf(v, 0.f);
                    C[(size_t)m * N + n] = f32_to_bf16(v);
                }
            }
        }
    }
}

extern "C" int mgx_linear_fwd(const uint16_t* A, const uint16_t* W, const float* bias, uint16_t* C, int M, int N,
                              int K, int act, void* stream) {
    MGX_REQUIRE(A && W && C, MGX_ERR_NULL, "mgx_linear_fwd: NULL pointer");
    MGX_REQUIRE(M > 0 && N > 0 && K > 0 && K % 64 == 0, MGX_ERR_SHAPE,
                "mgx_linear_fwd: need K%%64==0 (got M=%d N=%d K=%d)", M, N, K);
    MGX_REQUIRE(act == 0 || act == 1, MGX_ERR_SHAPE, "mgx_linear_fwd: act must be 0 (none) or 1 (ReLU)");
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)linear_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    const int nwg = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    hipLaunchKernelGGL(linear_fwd_kernel, dim3(nwg), dim3(256), LDS_BYTES, (hipStream_t)stream, A, W, bias, C, M, N,
                       K, act);
    MGX_CHECK_LAUNCH("mgx_linear_fwd");
    return MGX_OK;
}
